// stubs.hip -- entry points declared in mi355attn.h whose kernels are not built yet.  Each returns
// MI355_EUNSUPPORTED with a message (never a silent fallback).  Entries move out of this file as their
// kernels land; the file disappears when the header is fully implemented.
#include "common.h"
#define STUB(name) return mi355::fail(MI355_EUNSUPPORTED, name ": not implemented in this build")
extern "C" {
size_t mi355_double_attn_workspace_bytes(int, int, int, int, int, int) { return 16; }
int mi355_double_attn_fwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                          const float*, const float*, float*, int, int, int, int, int, int, int, void*, size_t,
                          mi355_stream_t) { STUB("mi355_double_attn_fwd"); }
int mi355_linear_fwd(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, int, int,
                     int, int, mi355_stream_t) { STUB("mi355_linear_fwd"); }
int mi355_token_mix_fwd(const float*, const float*, const float*, const float*, float*, int, int, int, int, int, int,
                        mi355_stream_t) { STUB("mi355_token_mix_fwd"); }
int mi355_layernorm_fwd(const float*, const float*, const float*, float*, int, int, float, mi355_stream_t) {
    STUB("mi355_layernorm_fwd"); }
int mi355_sdpa_fwd(const float*, float*, int, int, int, int, float, int, mi355_stream_t) { STUB("mi355_sdpa_fwd"); }
int mi355_cswin_lepe_attn_fwd(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int,
                              float, int, mi355_stream_t) { STUB("mi355_cswin_lepe_attn_fwd"); }
int mi355_xca_fwd(const float*, const float*, float*, int, int, int, int, int, mi355_stream_t) { STUB("mi355_xca_fwd"); }
size_t mi355_lpi_workspace_bytes(int, int, int, int) { return 16; }
int mi355_lpi_fwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float,
                  const float*, const float*, const float*, const float*, float*, int, int, int, int, void*, size_t,
                  mi355_stream_t) {
    STUB("mi355_lpi_fwd"); }
int mi355_patch_embed_fwd(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, int,
                          int, int, int, mi355_stream_t) { STUB("mi355_patch_embed_fwd"); }
}
