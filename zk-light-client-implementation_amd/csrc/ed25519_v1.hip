// Ed25519 verify kernel, tuning variant 1 (see ed25519_kernels.inc; selected at
// run time by zklc_ctx::ed_variant / env ZKLC_ED_VARIANT).
#define ZKLC_ED_VARIANT_ID 1
#define ZKLC_ED_BLOCK 64
#define ZKLC_ED_MINW 1
#define ZKLC_FE_INLINE 1
#include "ed25519_kernels.inc"
