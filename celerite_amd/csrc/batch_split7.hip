// celerite_amd/csrc/batch_split7.hip -- role-split summarize (clr_split_kernels.h), width 7,
// and the dispatcher over both widths.
#include "clr_split_kernels.h"

namespace clr {
bool launch_summarize_split_w8(const BatchParams& P, int JR, int JC, hipStream_t s);
static bool launch_summarize_split_w7(const BatchParams& P, int JR, int JC, hipStream_t s) {
  CLR_SPLIT_SHAPE(7, 0) CLR_SPLIT_SHAPE(5, 1) CLR_SPLIT_SHAPE(3, 2) CLR_SPLIT_SHAPE(1, 3)
  return false;
}
bool have_summarize_split(int JR, int JC) { const int J = JR + 2 * JC; return J == 7 || J == 8; }
bool launch_summarize_split(const BatchParams& P, int JR, int JC, hipStream_t s) {
  const int J = JR + 2 * JC;
  if (J == 8) return launch_summarize_split_w8(P, JR, JC, s);
  if (J == 7) return launch_summarize_split_w7(P, JR, JC, s);
  return false;
}
}  // namespace clr
