// celerite_amd/csrc/batch_w1.hip -- explicit instantiations of the batched scan
// kernels for width J = 1 (one translation unit per width so the fully
// unrolled kernels compile in parallel).  See clr_batch_kernels.h / clr_core.h.
#include "clr_batch_kernels.h"

namespace clr {
const BatchLaunchers* batch_launchers_w1(int JR, int JC) {
  if (JR == 1 && JC == 0) { static const BatchLaunchers L = BatchImpl<1, 0>::table(); return &L; }
  return nullptr;
}
}  // namespace clr
