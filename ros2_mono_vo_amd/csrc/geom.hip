// csrc/geom.hip — placeholder translation unit (RANSAC / PnP / triangulation land here).
#include "mvo_internal.h"
struct GeomState { int dummy; };
int geom_state_create(mvo_ctx* ctx) { ctx->geom = new GeomState(); return MVO_OK; }
void geom_state_destroy(mvo_ctx* ctx) { delete ctx->geom; ctx->geom = nullptr; }

int pipe_geometry_stages(mvo_ctx*, unsigned, mvo_step_result*) { return MVO_OK; }
