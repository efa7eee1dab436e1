// lab_hooks.h -- the seam between librmclhip.so (product) and librmclhip_lab.so (experiments).  The product's launchers own
// the traversal kinds the automatic rule can select and the round-3 particle-filter kernel; every other kind is handed to
// the hooks below, which the experiments' library registers when it is loaded (a static initialiser calls
// rmclhip_internal_register_lab).  Without that library those kinds report kLabMissing (kernels.h).  Not a public interface:
// include/rmclhip.h does not mention it; include/rmclhip_lab.h declares what tools/ and the `lab` tests may call.
#pragma once
#include "kernels.h"

namespace rmclhip {

struct LabHooks {
  // k_find kinds outside {0, 2, 23, 24, 32}; with_clock: the clocked instantiation of ANY kind (p.wave_clock != nullptr)
  hipError_t (*find)(const FindParams& p, ModelKind kind, int variant, bool with_clock, hipStream_t s);
  hipError_t (*find_probe)(const FindParams& p, int mode, uint32_t* probe_log, hipStream_t s);
  // particle filter: the round kernels (refill 0) and the round-2 persistent kernel
  hipError_t (*pf_update)(const PfParams& p, int variant, hipStream_t s);
};

// traversal kinds compiled into the product
constexpr bool find_kind_in_product(int variant) {
  return variant == 0 || variant == 2 || variant == 23 || variant == 24 || variant == 32;
}

const LabHooks* lab_hooks();   // nullptr until librmclhip_lab.so is loaded

}  // namespace rmclhip

extern "C" void rmclhip_internal_register_lab(const rmclhip::LabHooks* hooks);
