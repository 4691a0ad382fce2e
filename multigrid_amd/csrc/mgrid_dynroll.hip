// mgrid_dynroll.hip — the fused K-step rollout kernels of the mutable-grid kinds (mg_dyn_rollout_kernel, mg_dyn.h) as a code
// object of their own inside libmgrid.so.  The device code is mgrid.hip's, compiled a second time with MG_TU_DYNROLL: only
// those kernels and their lookup function are emitted here, none of the host API.  Why a second code object: see the note at
// DynVariant in mgrid.hip (38 more kernels in the main one cost the shared-map headline kernel 1.3 %).
#define MG_TU_DYNROLL 1
#include "mgrid.hip"
