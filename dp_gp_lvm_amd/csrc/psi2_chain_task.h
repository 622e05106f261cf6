// The K_uu branch of the ELBO as an extra task slice of a psi2 dispatch, and the one-dimensional item enumeration shared by
// the psi2 kernels that carry it (psi2.hip, psi2_pairs.hip).
#pragma once
#include "linalg_dev.h"

// Optional extra task slice in front of the psi2 workgroups (blockIdx.z == 0): the K_uu branch of the ELBO
// (chain_k_body of linalg_dev.h: Cholesky, log-det and inverse of K_uu for output dim blockIdx.x).  It is part of the SAME
// dispatch, ahead of the psi2 workgroups, because a separate dispatch on another stream is only served once the ~1500 psi2
// workgroups have all been placed (measured: it then finishes AFTER psi2 and lands on the critical path).
struct ChainKTask {
    void *ws;            // per-output Cholesky workspaces (float or double elements), nullptr = no task slice
    size_t ws_stride;
    double *logdet_k;
    int *info_k;
    int M, Mp, elem;     // elem = 4 (float) or 8 (double)
    int last;            // f16 kernel: tasks at the end of the grid instead of in front (see psi2_task_1d)
};
// OCC separates the instantiations by the launch bound of the calling kernel (the compiler derives the register budget of
// a device function from its callers; one shared copy would take the loosest bound and push the f16 kernel past 256 VGPRs)
template <int OCC>
__device__ __attribute__((always_inline, flatten)) void chain_k_task(const ChainKTask &tk, int d, unsigned char *smem_raw) {
    if (tk.elem == 8)
        chain_k_body<double, OCC>(d, tk.M, tk.Mp, (double *)tk.ws, tk.ws_stride, tk.logdet_k, tk.info_k, 0, smem_raw);
    else
        chain_k_body<float, OCC>(d, tk.M, tk.Mp, (float *)tk.ws, tk.ws_stride, tk.logdet_k, tk.info_k, 0, smem_raw);
}

// One-dimensional grid of the f16 kernel: C = B K_uu tasks (if fused) and P = B * ns * patches psi2 items (item j =
// b + B (sp + ns patch), long off-diagonal patches first).  The K_uu tasks are latency bound (one small Cholesky each, hardly
// any VALU work) and a psi2 workgroup needs a second psi2 workgroup on its compute unit to keep the vector units busy (one
// wave per SIMD reaches ~2/3 of the issue rate).  Measured placements of the K_uu tasks (config 3 / config 2, evals/s):
//   en bloc in front 515 / 2499;  en bloc at the end 545 / 2064;  interleaved with psi2 items in runs of 8: 457 / 2452
//   (next to a psi2 workgroup a K_uu task takes ~600 us instead of ~100 us, whatever its s_setprio).
// In front, B >= ~256 tasks hold every slot of the GPU for ~100 us with idle vector units; at the end they fill the slots
// the psi2 tail leaves empty anyway, but add their full latency when there are only few of them.  Hence: at the end iff
// B >= 256.
__device__ __forceinline__ bool psi2_task_1d(int id, int C, bool chain_last, int &task) {
    if (chain_last) {
        const int P = (int)gridDim.x - C;
        if (id < P) { task = id; return false; }
        task = id - P;
        return true;
    }
    if (id < C) { task = id; return true; }
    task = id - C;
    return false;
}

