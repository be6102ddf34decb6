// A-DISTS kernels (nerf_qa/ADISTS/ADISTS.py:71-197).  Placeholder until the windowed
// statistics kernels land: the entry points exist so the ABI is complete and fail loudly.
#include "nqa_common.h"

using namespace nqa;

extern "C" {

size_t nqa_adists_workspace_bytes(int B, int H, int W, int prec) {
  (void)B; (void)H; (void)W; (void)prec;
  return 0;
}

int nqa_adists_forward(const float *, const float *, int, int, int, const void *, int, void *, size_t, float *,
                       void *) {
  set_error("adists_forward: not implemented in this build");
  return NQA_E_SHAPE;
}

}  // extern "C"
