// TEST INFRASTRUCTURE ONLY: the per-element arithmetic of xrnerf_amd/csrc/xr_mip.hip (xr_mip_math.h, the very header
// the kernels include) compiled for the host, so that tests/test_mip_host_math.py can hold it against the numpy
// oracle without a GPU.  Nothing in xrnerf_amd/ loads this; the product path is the HIP library only.
#include "../../xrnerf_amd/csrc/xr_mip_math.h"

extern "C" void hm_zvals(const float* near, const float* far, unsigned n_rays, unsigned n_z, int lindisp, float* z) {
    for (unsigned r = 0; r < n_rays; ++r)
        for (unsigned j = 0; j < n_z; ++j) z[r * n_z + j] = xr_mip_zval(near[r], far[r], n_z, j, lindisp);
}

extern "C" void hm_encode(const float* o, const float* d, const float* vd, const float* radii, const float* z,
                          unsigned n_rays, unsigned n_z, int min_deg, int max_deg, int min_deg_view, int max_deg_view,
                          int append_identity, int cylinder, unsigned ch, float* out) {
    const unsigned S = n_z - 1;
    for (unsigned r = 0; r < n_rays; ++r)
        for (unsigned s = 0; s < S; ++s) {
            float mean[3], cov[3];
            xr_mip_gaussian(o + 3 * r, d + 3 * r, radii[r], z[r * n_z + s], z[r * n_z + s + 1], cylinder, mean, cov);
            for (unsigned c = 0; c < ch; ++c)
                out[((size_t)r * S + s) * ch + c] =
                    xr_mip_feature(c, mean, cov, vd + 3 * r, min_deg, max_deg, min_deg_view, max_deg_view, append_identity);
        }
}

extern "C" void hm_linspace(float start, float end, unsigned n, float* out) {
    for (unsigned j = 0; j < n; ++j) out[j] = xr_torch_linspace(start, end, n, j);
}

extern "C" void hm_density(const float* x, unsigned n, int relu, float* act, float* dact, float* sig) {
    for (unsigned i = 0; i < n; ++i) {
        act[i] = xr_mip_density_act(x[i], relu);
        dact[i] = xr_mip_density_dact(x[i], relu);
        sig[i] = xr_mip_sigmoid(x[i]);
    }
}
