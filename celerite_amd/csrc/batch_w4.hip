// celerite_amd/csrc/batch_w4.hip -- explicit instantiations of the batched scan
// kernels for width J = 4 (one translation unit per width so the fully
// unrolled kernels compile in parallel).  See clr_batch_kernels.h / clr_core.h.
#include "clr_batch_kernels.h"

namespace clr {
const BatchLaunchers* batch_launchers_w4(int JR, int JC) {
  if (JR == 4 && JC == 0) { static const BatchLaunchers L = BatchImpl<4, 0>::table(); return &L; }
  if (JR == 2 && JC == 1) { static const BatchLaunchers L = BatchImpl<2, 1>::table(); return &L; }
  if (JR == 0 && JC == 2) { static const BatchLaunchers L = BatchImpl<0, 2>::table(); return &L; }
  return nullptr;
}
}  // namespace clr
