// Device-side target log-densities evaluated on the 16-chain tile of a workgroup.
// ManyWell: fab/target_distributions/many_well.py:81-90 -> double_well.py:44-58
// GMM:      fab/target_distributions/gmm.py:57-66 (MixtureSameFamily of diagonal normals, -inf mask)
#pragma once
#include "flow_device.h"

namespace fab {

struct TargetDev {
    int kind, dim;
    float a, b, c, log_norm;
    int n_mix;
    const float* locs;
    const float* scales;
};

static inline TargetDev make_target_dev(const fabhip_target& t) {
    TargetDev d;
    d.kind = t.kind; d.dim = t.dim; d.a = t.a; d.b = t.b; d.c = t.c; d.log_norm = t.log_norm;
    d.n_mix = t.n_mix; d.locs = t.locs; d.scales = t.scales;
    return d;
}

static inline int check_target(const fabhip_target* t, int dim) {
    if (!t) return FABHIP_EINVAL;
    if (t->dim != dim) return FABHIP_EINVAL;
    if (t->kind == FABHIP_TARGET_MANYWELL) return (dim % 2 == 0) ? FABHIP_OK : FABHIP_EINVAL;
    if (t->kind == FABHIP_TARGET_GMM) return (t->n_mix > 0 && t->locs && t->scales) ? FABHIP_OK : FABHIP_EINVAL;
    return FABHIP_ENOTSUP;
}

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, row_ror<8>(v));
    v = fmaxf(v, row_ror<4>(v));
    v = fmaxf(v, row_ror<2>(v));
    v = fmaxf(v, row_ror<1>(v));
    return v;
}

// X: LDS [16][ldx] positions; GP: LDS [16][ldg] receives d log p / dx when GRAD.
// Returns log p of this thread's row (replicated over its 16 lanes).
template <bool GRAD>
__device__ float target_tile(const TargetDev& tg, const float* X, int ldx, float* GP, int ldg, const Tid& t) {
    const int D = tg.dim;
    if (tg.kind == FABHIP_TARGET_MANYWELL) {
        float acc = 0.f;
        for (int j = t.c; j < D; j += 16) {
            const float x = X[t.row * ldx + j];
            const float x2 = x * x;
            float e, g;
            if ((j & 1) == 0) {
                e = tg.a * x + tg.b * x2 + tg.c * (x2 * x2);
                g = -(tg.a + 2.f * tg.b * x + 4.f * tg.c * (x2 * x));
            } else {
                e = 0.5f * x2;
                g = -x;
            }
            acc += -e;
            if (GRAD) GP[t.row * ldg + j] = g;
        }
        return row16_sum(acc) - tg.log_norm;
    }
    // ---- GMM: lane c handles components k = c, c+16, ... ----------------------------------------
    const float LOG2PI = 1.8378770664093453f;
    const float log_mix = -logf((float)tg.n_mix);
    float best = -INFINITY;
    bool any_nan = false;
    for (int k = t.c; k < tg.n_mix; k += 16) {
        float m = 0.f, hld = 0.f;
        for (int j = 0; j < D; ++j) {
            const float sc = tg.scales[k * D + j];
            const float z = (X[t.row * ldx + j] - tg.locs[k * D + j]) / sc;
            m += z * z;
            hld += logf(sc);
        }
        const float comp = -0.5f * ((float)D * LOG2PI + m) - hld + log_mix;
        any_nan |= (comp != comp);
        best = fmaxf(best, comp);
    }
    const float mx = row16_max(best);
    float se = 0.f;
    for (int k = t.c; k < tg.n_mix; k += 16) {
        float m = 0.f, hld = 0.f;
        for (int j = 0; j < D; ++j) {
            const float sc = tg.scales[k * D + j];
            const float z = (X[t.row * ldx + j] - tg.locs[k * D + j]) / sc;
            m += z * z;
            hld += logf(sc);
        }
        const float comp = -0.5f * ((float)D * LOG2PI + m) - hld + log_mix;
        se += (mx == -INFINITY) ? 0.f : expf(comp - mx);
    }
    const float stot = row16_sum(se);
    const float nanflag = row16_sum(any_nan ? 1.f : 0.f);
    float lp = (mx == -INFINITY) ? -INFINITY : mx + logf(stot);
    if (nanflag > 0.f) lp = NAN;
    if (GRAD) {
        for (int j = 0; j < D; ++j) {
            float gj = 0.f;
            for (int k = t.c; k < tg.n_mix; k += 16) {
                float m = 0.f, hld = 0.f;
                for (int jj = 0; jj < D; ++jj) {
                    const float sc = tg.scales[k * D + jj];
                    const float z = (X[t.row * ldx + jj] - tg.locs[k * D + jj]) / sc;
                    m += z * z;
                    hld += logf(sc);
                }
                const float comp = -0.5f * ((float)D * LOG2PI + m) - hld + log_mix;
                const float w = expf(comp - lp);
                const float sc = tg.scales[k * D + j];
                gj += w * (-(X[t.row * ldx + j] - tg.locs[k * D + j]) / (sc * sc));
            }
            gj = row16_sum(gj);
            GP[t.row * ldg + j] = gj;          // all 16 lanes of the row store the same value
        }
    }
    if (lp < -1e4f) lp = -INFINITY;       // gmm.py:63-65
    return lp - tg.log_norm;
}

}  // namespace fab
