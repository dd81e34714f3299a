// celerite_amd/csrc/batch_split8.hip -- role-split summarize (clr_split_kernels.h), width 8.
#include "clr_split_kernels.h"

namespace clr {
bool launch_summarize_split_w8(const BatchParams& P, int JR, int JC, hipStream_t s) {
  CLR_SPLIT_SHAPE(8, 0) CLR_SPLIT_SHAPE(6, 1) CLR_SPLIT_SHAPE(4, 2) CLR_SPLIT_SHAPE(2, 3) CLR_SPLIT_SHAPE(0, 4)
  return false;
}
}  // namespace clr
