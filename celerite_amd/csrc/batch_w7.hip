// celerite_amd/csrc/batch_w7.hip -- explicit instantiations of the batched scan
// kernels for width J = 7 (one translation unit per width so the fully
// unrolled kernels compile in parallel).  See clr_batch_kernels.h / clr_core.h.
#include "clr_batch_kernels.h"

namespace clr {
const BatchLaunchers* batch_launchers_w7(int JR, int JC) {
  if (JR == 7 && JC == 0) { static const BatchLaunchers L = BatchImpl<7, 0>::table(); return &L; }
  if (JR == 5 && JC == 1) { static const BatchLaunchers L = BatchImpl<5, 1>::table(); return &L; }
  if (JR == 3 && JC == 2) { static const BatchLaunchers L = BatchImpl<3, 2>::table(); return &L; }
  if (JR == 1 && JC == 3) { static const BatchLaunchers L = BatchImpl<1, 3>::table(); return &L; }
  return nullptr;
}
}  // namespace clr
