// library-internal interface of the third-generation scatter (xr_scatter.hip), called from xr_hashgrid_bwd2 (xr_encode.hip)
#pragma once
#include "xr_hashgrid.h"
struct XrAdamArgs;

// bytes of workspace xr_scatter3 needs for n rows (0: this n / geometry takes the atomic kernel for every level)
size_t xr_scatter3_workspace_bytes(uint32_t n, const GridMeta& gm, uint32_t hashed_mask);
// the levels xr_scatter3 leaves to the atomic kernel for this n / geometry (all of them without a usable workspace)
uint32_t xr_scatter3_atomic_mask(uint32_t n, const GridMeta& gm, uint32_t hashed_mask, bool workspace_ok);
// scatters every level it has a non-atomic path for and reports the others in *atomic_mask (the caller runs the atomic
// kernel on those).  overwrite != 0: the levels' table slices are written, not added to.
// adam (nullable): the optimiser update is applied to every entry of the levels instead of writing their gradient (grad_table
// is then only checked for alignment); an error unless every level has a non-atomic path.
int xr_scatter3(const float* x, uint32_t x_stride, const float* denc_t, uint32_t ld, uint32_t n, const uint32_t* n_dev,
                const uint32_t* rows, const GridMeta& gm, uint32_t hashed_mask, float* grad_table, void* workspace,
                size_t workspace_bytes, int overwrite, uint32_t* atomic_mask, hipStream_t stream, const struct XrAdamArgs* adam = nullptr);
