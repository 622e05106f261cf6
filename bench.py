#!/usr/bin/env python3
"""
Benchmark of the DP-GP-LVM objective (ELBO) evaluation — BASELINE.json metric "ELBO evals/sec (N=2000,D=512,M=128,Q=10)".

One step = one evaluation of dp_gp_lvm(...).objective (reference: src/models/dp_gp_lvm.py:100-154) from device-resident
raw parameters to the scalar objective in device memory: parameter transforms + soft-assignment mixing + DP objective +
hyper-prior (dpgp_model_prepare), K_uu, Psi1^T y, Psi2, both Choleskys + solves + the five f_hat terms, KL
(dpgp_elbo_fhat), the packed 2-scalar all-reduce when D is sharded over GPUs (RCCL), and dpgp_model_finalize.
Steps are enqueued back to back and the host synchronises once at the end of the timed region (what a training loop
does: the reference reads the objective only every 100 iterations, test/synthetic_data_hard_test.py:143-155); every
step's objective is kept on the device and checked after timing.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--prec mixed] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `value` is whole-job evaluations/s (D=512 is a fixed total: strong scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_F32_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak (spec), gfx950
MFMA_F16_PEAK_TFLOPS = 2516.6    # same guide: ~2.5 PF dense f16/bf16 (256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', type=int, default=3, help='BASELINE.json config index (SURVEY.md 8d): 2,3,4,5')
    ap.add_argument('--prec', default='mixed', choices=['mixed', 'f32', 'f64'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the gram GB/s and Cholesky TFLOP/s side measurements')
    ap.add_argument('--no-grad', action='store_true', help='skip the objective + gradients side measurement (first-version backward pass)')
    ap.add_argument('--cpu-dims', type=int, default=0, help='output dims in the bounded CPU sample (0 = auto)')
    return ap.parse_args()


def cpu_baseline(cfg, p, shape, dims):
    """The C restatement of the reference algorithm (oracle/dpgp_oracle.c, fast build, OpenMP over output dims) timed on
    this box's host cores on a bounded sample: `dims` of the D output dims (every dim costs the same), 3 repetitions."""
    from oracle.c_oracle import COracle
    n, d, m, q = shape
    orc = COracle(fast=True)
    cores = orc.max_threads
    dims = min(d, dims if dims > 0 else max(2 * cores, 16))
    sel = np.linspace(0, d - 1, dims).astype(int)
    args = (np.ascontiguousarray(p['y'][:, sel]), p['z'], p['mu'], p['s'], p['gamma'][sel], p['alpha'][sel], p['beta'][sel])
    orc.fhat_terms(*args, nthreads=cores)                    # warm-up (page in, spin up the OpenMP pool)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        orc.fhat_terms(*args, nthreads=cores)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    return dict(value=1.0 / (t * d / dims), unit='ELBO evals/s', cores=int(cores), kind='port',
                sample='%d of %d output dims of config %d, all of N=%d M=%d Q=%d, fp64 C/OpenMP port of the reference '
                       'formulas (oracle/dpgp_oracle.c), median of 3 runs of %.2f s, scaled by D/dims'
                       % (dims, d, cfg, n, m, q, t))


def secondary(dev, shape, p):
    """The two kernel-level figures BASELINE.json names next to the headline metric, measured on their own (torch events on
    the current stream, the one the operators launch on): HBM GB/s of the gram build (K_uu, fp64 as the mixed pipeline
    writes it, plus one large fp32 gram that is not launch-bound) and TFLOP/s of the batched Cholesky (M^3/3 flops per
    factorisation; fp64 as in the mixed pipeline; the workload's M and M = 512, the blocked-MFMA regime of config 4)."""
    import torch
    from dp_gp_lvm_amd import ops
    n, d, m, q = shape
    out = {}

    def timed(fn, reps, setup=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for i in range(reps + 2):
            if setup:
                setup()
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            if i >= 2:
                tot += e0.elapsed_time(e1)
        return tot / reps

    t64 = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64, device=dev)
    z, g, al, be = t64(p['z']), t64(p['gamma']), t64(p['alpha']), t64(p['beta'])
    ms = timed(lambda: ops.ard_rbf_gram(z, None, g, al, be, include_jitter=True), 20)
    byts = 8.0 * d * m * m
    out['gram_kuu'] = {'kernel_ms': ms, 'bytes': byts, 'gbps': byts / ms / 1e6, 'peak_gbps': 8000.0,
                       'frac': byts / ms / 1e6 / 8000.0, 'what': 'K_uu [D=%d,M=%d,M=%d] fp64 incl. operator call overhead' % (d, m, m)}
    rng = np.random.default_rng(7)
    x = torch.as_tensor(rng.standard_normal((4096, q)), dtype=torch.float32, device=dev)
    g32, a32, b32 = g[:16].float().contiguous(), al[:16].float().contiguous(), be[:16].float().contiguous()
    ms = timed(lambda: ops.ard_rbf_gram(x, None, g32, a32, b32), 10)
    byts = 4.0 * 16 * 4096 * 4096
    out['gram_large'] = {'kernel_ms': ms, 'bytes': byts, 'gbps': byts / ms / 1e6, 'peak_gbps': 8000.0,
                         'frac': byts / ms / 1e6 / 8000.0, 'what': 'gram [16,4096,4096] fp32'}
    for key, (bb, mm) in {'cholesky': (d, m), 'cholesky_m512': (256, 512)}.items():     # (config 4: D = 256, M = 512)
        a0 = torch.as_tensor(rng.standard_normal((bb, mm, mm)), dtype=torch.float64, device=dev)
        spd = a0 @ a0.transpose(1, 2) + mm * torch.eye(mm, dtype=torch.float64, device=dev)
        ms = timed(lambda: ops.potrf_batched(spd), 10)       # the operator factorises a copy of its argument
        ms_copy = timed(lambda: spd.clone(), 10)
        ms = max(ms - ms_copy, 1e-6)
        fl = bb * mm ** 3 / 3.0
        out[key] = {'kernel_ms': ms, 'flops': fl, 'tflops': fl / ms / 1e9, 'peak_tflops': 78.6, 'frac': fl / ms / 1e9 / 78.6,
                    'what': 'dpgp_potrf_batched_f64 B=%d M=%d (M^3/3 flops each), copy of the input subtracted' % (bb, mm)}
    return out


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from dp_gp_lvm_amd import _lib
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (a.gpus, world, a.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    group = None
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)      # nccl == RCCL on ROCm
        group = dist.group.WORLD

    shape = CONFIGS[a.config]
    n, d, m, q = shape
    p = make_problem(a.config)
    t = p['phi'].shape[1]
    init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'],
                alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'],
                w_2=p['w2'])
    model = dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=t,
                      alpha_prior_params=np.array([p['s1'], p['s2']]), device=dev, precision=a.prec,
                      process_group=group, initial_values=init)
    lib = _lib.lib()
    d_lo, d_hi = model.shard
    outs = torch.zeros((a.steps, 5), dtype=torch.float64, device=dev)   # every step's objective breakdown stays on the device
    # HIP events around the psi2 kernel on every EV_EVERY-th step only: each recorded event costs ~6 us of stream time
    EV_EVERY = 8
    ev = {i: (lib.dpgp_event_create(), lib.dpgp_event_create()) for i in range(0, a.steps, EV_EVERY)}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        model.evaluate_()
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        model.evaluate_(events=ev.get(i), out=outs[i])
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    psi2_ms = float(np.mean([lib.dpgp_event_elapsed_ms(e0, e1) for e0, e1 in ev.values()]))
    for e0, e1 in ev.values():
        lib.dpgp_event_destroy(e0)
        lib.dpgp_event_destroy(e1)
    objs = outs[:, 0].cpu().numpy()
    terms, info = model.per_dimension_terms
    assert np.isfinite(objs).all() and np.all(objs == objs[0]), 'objective is not finite / not reproducible'
    assert int(info.abs().max().item()) == 0, 'a Cholesky factorisation failed'

    if rank == 0:
        d_loc = d_hi - d_lo
        # algorithmic work of the dominant kernel (psi2) per launch: SURVEY.md 8(d): per output dim N*M(M+1)/2 exps and
        # (4Q+2) flops per exponent on the symmetric half; one launch processes d_loc output dims.
        exps = d_loc * n * m * (m + 1) // 2
        flops = exps * (4 * q + 2)
        achieved = flops / (psi2_ms * 1e-3) / 1e12
        # HBM traffic of one psi2 dispatch (incl. its K_uu task slice, now LDS-resident) from the rocprofv3 PMC passes
        # committed under profiles/r01 (FETCH_SIZE 25318 + WRITE_SIZE 67616, KiB -> bytes; config 3, mixed precision,
        # 1 GPU); null for every other configuration (not profiled)
        traffic = 95.2e6 if (a.config == 3 and a.prec == 'mixed' and world == 1) else None
        exp_peak = 256 * 4 * 8 * 2.4e9        # v_exp_f32: 64 lanes / 8 cycles per SIMD, 1024 SIMDs, 2.4 GHz
        # measured issue floor of the hot loop (profiles/r01/ubench_psi2_row_mix_floor.txt: the per-row instruction mix of
        # a 64 x 64 patch -- 64 v_exp_f32, 32 v_pk_add_f32, 12 MFMA 32x32x16 f16, 30 split, 16 v_mov -- issues in 1209
        # cycles per row and SIMD with two waves per SIMD, Q <= 12): rows x patches (a diagonal patch has 3 of 4 tiles)
        floor_ms = None
        if q <= 12 and a.prec != 'f64':
            np64 = (m + 63) // 64
            tile_rows = d_loc * n * (np64 * (np64 - 1) // 2 * 1.0 + np64 * 0.75)
            floor_ms = tile_rows * 1209.0 / (256 * 4) / 2.4e9 * 1e3
        f16_path = a.prec != 'f64'
        peak = MFMA_F16_PEAK_TFLOPS if f16_path else 78.6
        res = {
            'metric': 'ELBO evals/sec (N=%d,D=%d,M=%d,Q=%d)' % (n, d, m, q),
            'value': a.steps / elapsed, 'unit': 'ELBO evals/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * elapsed / a.steps, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': {'mixed': 'f32 (psi-statistics: f16 hi/lo-split MFMA operands, fp32 accumulate) + '
                                                    'f64 (Cholesky chain)', 'f32': 'f32', 'f64': 'f64'}[a.prec],
            'data': 'synthetic (SURVEY.md 8d recipe, seed %d)' % (1000 + a.config),
            'config': {'workload': 'BASELINE config %d: dp_gp_lvm objective, N=%d D=%d M=%d Q=%d T=%d' % (a.config, n, d, m, q, t),
                       'parallelism': 'D sharded over %d GPU(s), %d output dims per GPU' % (world, d_loc),
                       'precision': a.prec},
            'objective': float(objs[0]),
            # dominant kernel: psi2 (+ the K_uu task slice that rides in the same dispatch).  `achieved` = ALGORITHMIC flops
            # (N M(M+1)/2 exponents x (4Q+2) flops per output dim) / HIP-event duration; `peak` = dense peak of
            # the matrix pipe the kernel runs on (f16 operands, fp32 accumulate; fp64 MFMA for --prec f64).  The matrix pipe
            # is not what limits it: every exponent costs one v_exp_f32 (8 issue cycles per wave) + one accumulate, see
            # `exp_frac` (vs the v_exp_f32 rate alone) and `issue_floor_frac` (vs the measured issue rate of the loop's
            # whole instruction mix) and DESIGN.md section 4.
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic,
                         'kernel': 'psi2_f16_kernel' if f16_path else 'psi2_mfma_kernel', 'kernel_ms': psi2_ms,
                         'limiter': 'VALU issue (v_exp_f32 + accumulate + operand split), not the matrix pipe',
                         'vs_fp32_matrix_peak': achieved / MFMA_F32_PEAK_TFLOPS,
                         'exp_per_s': exps / (psi2_ms * 1e-3), 'exp_peak_per_s': exp_peak,
                         'exp_frac': exps / (psi2_ms * 1e-3) / exp_peak,
                         'issue_floor_ms': floor_ms, 'issue_floor_frac': (floor_ms / psi2_ms) if floor_ms else None},
        }
        if world == 1 and not a.no_secondary:
            res['secondary'] = secondary(dev, shape, p)
        if world == 1 and not a.no_grad and (a.prec == 'mixed' or (a.prec == 'f64' and m <= 128)):
            # side measurements, not the headline metric: (1) one objective evaluation + the gradients of all raw variables
            # (backward pass; what one Adam iteration of the reference needs); (2) one objective evaluation
            # of the over-T formulation dp_gp_lvm_t on the same data (8f row 3: T Psi2's instead of D)
            for _ in range(2):
                model.gradients()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                model.gradients()
            torch.cuda.synchronize()
            res['objective_and_gradients'] = {'ms': 1e3 * (time.perf_counter() - t0) / reps, 'reps': reps,
                                              'note': 'stage B of the backward pass on the matrix pipe in mixed precision '
                                                      '(psi2_grad_kernel), plain kernel in f64; DESIGN.md 7.1'}
            # breakdown of one more iteration (torch events on the launch stream): forward, stage A, stage B, chain rule; and
            # the exponential rate of stage B (its Psi2 term evaluates the FULL M x M square of every (d, n))
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            evs[4].record()
            model.gradients(events=evs[:4])
            torch.cuda.synchronize()
            t_fwd, t_a, t_b, t_c = (evs[4].elapsed_time(evs[0]), evs[0].elapsed_time(evs[1]), evs[1].elapsed_time(evs[2]),
                                    evs[2].elapsed_time(evs[3]))
            mp64 = 64 * ((m + 63) // 64)
            exps_b = float(n) * (d_hi - d_lo) * mp64 * mp64
            res['objective_and_gradients'].update({
                'forward_ms': t_fwd, 'stage_a_ms': t_a, 'stage_b_ms': t_b, 'chain_rule_ms': t_c,
                'stage_b_exp_per_s': exps_b / (t_b * 1e-3), 'stage_b_exp_frac_of_v_exp_rate': exps_b / (t_b * 1e-3) / exp_peak})
            if a.prec == 'mixed' and m <= 128:
                # what optimise() falls back to when fp32 Psi2 is no longer accurate enough: fp64 forward
                # and dense adjoints, streaming stage B on the matrix pipe
                tw = model.fp64_twin()
                for _ in range(2):
                    tw.gradients()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    tw.gradients()
                torch.cuda.synchronize()
                res['objective_and_gradients']['fp64_forward_matrix_pipe_stage_b_ms'] = 1e3 * (time.perf_counter() - t0) / 5
            if a.prec in ('mixed', 'f64'):
                from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm_t
                model_t = dp_gp_lvm_t(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=p['phi'].shape[1],
                                      alpha_prior_params=np.array([p['s1'], p['s2']]), device=dev, precision=a.prec,
                                      initial_values=init)
                for _ in range(2):
                    model_t.objective_terms
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    model_t.objective_terms
                torch.cuda.synchronize()
                res['objective_over_t'] = {'ms': 1e3 * (time.perf_counter() - t0) / reps, 'reps': reps,
                                           'truncation_level': int(p['phi'].shape[1]),
                                           'note': 'dp_gp_lvm_t objective, composed of the library operators (not fused)'}
                if a.prec == 'mixed':
                    for _ in range(2):
                        model_t.gradients()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        model_t.gradients()
                    torch.cuda.synchronize()
                    res['objective_over_t']['with_gradients_ms'] = 1e3 * (time.perf_counter() - t0) / reps
        if not a.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline(a.config, p, shape, a.cpu_dims)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
