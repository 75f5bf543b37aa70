// score.hip — placeholder while the scoring kernels are being written (replaced in the next commit).
#include "index.hpp"
struct ss_scorer { ss_ctx* ctx; };
extern "C" {
int32_t ss_scorer_create(ss_ctx* ctx, ss_index*, ss_index*, ss_scorer**) { return ctx ? ctx->fail(SS_ERR_UNSUPPORTED, "scorer not built yet") : SS_ERR_INVALID; }
int32_t ss_scorer_destroy(ss_scorer*) { return SS_ERR_INVALID; }
int32_t ss_scorer_set_prior(ss_scorer*, int32_t, const double*) { return SS_ERR_INVALID; }
int32_t ss_score_topk(ss_scorer*, int32_t, const uint32_t*, const uint32_t*, const int32_t*, const double*, int32_t, ss_hit*, int32_t*) { return SS_ERR_INVALID; }
}
