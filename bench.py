#!/usr/bin/env python3
"""
Benchmark of the DP-GP-LVM objective (ELBO) evaluation — BASELINE.json metric "ELBO evals/sec (N=2000,D=512,M=128,Q=10)".

One step = one evaluation of dp_gp_lvm(...).objective (reference: src/models/dp_gp_lvm.py:100-154) from device-resident
raw parameters to the scalar objective in device memory: parameter transforms + soft-assignment mixing + DP objective +
hyper-prior (dpgp_model_prepare), K_uu, Psi1^T y, Psi2, both Choleskys + solves + the five f_hat terms, KL
(dpgp_elbo_fhat), the packed 2-scalar all-reduce when D is sharded over GPUs (RCCL), and dpgp_model_finalize.
Every step ends with the objective copied to pinned host memory on the stream (host-visible per evaluation); steps are
enqueued back to back and the host synchronises once at the end of the timed region (what a training loop does: the
reference reads the objective only every 100 iterations, test/synthetic_data_hard_test.py:143-155); every step's value is
checked after timing.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--prec mixed] [--graph auto|on|off] [--no-cpu-baseline]
        (N > 1 without a rendezvous in the environment: starts the N ranks itself as a child torch.distributed.run)
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
    ap.add_argument('--no-side', action='store_true', help='skip the exact-fp32 and fp64 side runs of the same metric')
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='replay the evaluation from a HIP graph (auto = off: measured no faster)')
    ap.add_argument('--cpu-dims', type=int, default=0, help='output dims in the bounded CPU sample (0 = auto)')
    return ap.parse_args()


def host_cpu():
    """(model string, usable cores): the box's CPU share — the scheduler affinity capped by the cgroup quota (a GPU box gives
    one GPU's worker a slice of the host, os.cpu_count() is the whole machine)."""
    model = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return model, cores


def cpu_baseline(cfg, p, shape, dims):
    """SURVEY.md 8(d) protocol.  The C restatement of the reference algorithm (oracle/dpgp_oracle.c, -O3 -ffast-math build with
    libmvec's vector exp: AVX-512 when the host has it, else AVX2; OpenMP over output dims) timed on this box's host cores on a
    bounded sample: `dims` of the D output dims (every dim costs the same), 2 warm-ups, median of 5 repetitions, on all usable
    cores and on ONE thread.  The reference as written cannot hold configs 2-5 (168 GB ... 107 PB temporaries);
    its own source under the NumPy stand-in at config 1 takes 1.25 s per objective build (quoted)."""
    from oracle.c_oracle import COracle
    n, d, m, q = shape
    orc = COracle(fast=True)
    model, cores = host_cpu()
    cores = max(1, min(cores, orc.max_threads))

    def timed(sel, threads):
        args = (np.ascontiguousarray(p['y'][:, sel]), p['z'], p['mu'], p['s'], p['gamma'][sel], p['alpha'][sel], p['beta'][sel])
        for _ in range(2):
            orc.fhat_terms(*args, nthreads=threads)          # warm-ups (page in, spin up the OpenMP pool)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            orc.fhat_terms(*args, nthreads=threads)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))
    dims = min(d, dims if dims > 0 else max(2 * cores, 16))
    sel = np.linspace(0, d - 1, dims).astype(int)
    t_all = timed(sel, cores)
    dims1 = max(1, min(dims, 1 if n * m * m * q > 1e10 else 8))
    sel1 = np.linspace(0, d - 1, dims1).astype(int)
    t_one = timed(sel1, 1)
    return dict(value=1.0 / (t_all * d / dims), unit='ELBO evals/s', cores=int(cores), kind='port',
                sample='%d of %d output dims of config %d, all of N=%d M=%d Q=%d; fp64 C/OpenMP port of the reference formulas '
                       '(oracle/dpgp_oracle.c, %s: vector exp), 2 warm-ups + median of 5 runs of %.2f s, scaled by D/dims'
                       % (dims, d, cfg, n, m, q, orc.build_name, t_all),
                cpu_model=model, host_logical_cpus=int(os.cpu_count() or 0),
                single_thread={'value': 1.0 / (t_one * d / dims1), 'unit': 'ELBO evals/s',
                               'sample': '%d output dims, median of 5 runs of %.2f s' % (dims1, t_one)},
                reference_as_written={'config': 'BASELINE config 1 (N=100, D=12, M=20, Q=4)', 'seconds_per_objective_build': 1.25,
                                      'what': "the reference's own source under the NumPy stand-in for TensorFlow, build "
                                              'container (8 cores); quoted from BASELINE.md section 2, not re-measured here'})


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
        """ms per call: events around `reps` back-to-back calls (the queue stays full, so this is device time, not the
        host-side cost of one operator call — round 2 timed single calls and reported the 0.2 ms gram kernel as 0.33 ms)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

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
    # counter-derived figures of the same workloads (scratch/prof_r04.sh: WRITE_SIZE / FETCH_SIZE in separate --pmc passes,
    # SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flops, kernel-trace durations), committed with the build they were taken on
    try:
        rnd = 'r04' if os.path.exists(os.path.join(REPO, 'profiles', 'r04', 'linalg_pmc.json')) else 'r03'
        pmc = json.load(open(os.path.join(REPO, 'profiles', rnd, 'linalg_pmc.json')))
        sha = open(os.path.join(REPO, 'profiles', rnd, 'GIT_SHA_OF_PROFILED_BUILD.txt')).read().split()[0]
        pick = lambda k: {a: b for a, b in pmc[k].items() if a != 'counters'} if k in pmc else None
        out['rocprof'] = {'source': 'profiles/%s/linalg_pmc.json @ ' % rnd + sha,
                          'gram_kuu': pick('void gram_kernel<double, double>'), 'gram_large': pick('void gram_kernel<float, float>'),
                          'cholesky_m128_b512': pick('void potrf_batched_lds_kernel<double>'),
                          'cholesky_m512_b256': pick('pleft_persistent_kernel') or pick('pbig_persistent_kernel')}
    except (OSError, ValueError, KeyError):
        pass
    return out


def git_sha():
    try:
        import subprocess
        return subprocess.check_output(['git', '-C', REPO, 'rev-parse', '--short=12', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:                                            # the GPU box receives the tree without .git
        return None


def profiled_traffic(cfg, prec, world):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): read from
    profiles/r04/traffic.json (older rounds as fall-back), which records the build it was measured on; null when this configuration
    was not profiled."""
    path = os.path.join(REPO, 'profiles', 'r04', 'traffic.json')
    for older in ('r03', 'r02'):
        if not os.path.exists(path):
            path = os.path.join(REPO, 'profiles', older, 'traffic.json')
    if world != 1 or not os.path.exists(path):
        return None, None
    try:
        rec = json.load(open(path)).get('config%d_%s' % (cfg, prec))
    except (OSError, ValueError):
        return None, None
    if not rec:
        return None, None
    return float(rec['bytes_per_launch']), rec


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment (the driver's command line): start the N
    ranks as a CHILD process group — `python -m torch.distributed.run --nnodes 1 --nproc-per-node N bench.py <same flags>` —
    before this process has imported torch or touched the GPU, let rank 0's JSON line through on stdout and return the
    child's exit code.  (Never os.exec*: replacing a process that has initialised the GPU takes the machine down.)"""
    import socket
    import subprocess
    port = os.environ.get('MASTER_PORT')
    if not port:
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = str(sock.getsockname()[1])
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(a))
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
    # DPGP_BENCH_BACKEND=gloo (rehearsal only): the world > 1 code path with all ranks on the visible device(s) — RCCL refuses
    # two ranks on one GPU, the development box has one.  Numbers of such a run are not benchmark results.
    backend = os.environ.get('DPGP_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    group = None
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    shape = CONFIGS[a.config]
    n, d, m, q = shape
    p = make_problem(a.config)
    t = p['phi'].shape[1]
    init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'],
                alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'],
                w_2=p['w2'])
    kw = dict(num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]),
              device=dev, initial_values=init)
    model = dp_gp_lvm(p['y'], precision=a.prec, process_group=group, **kw)
    lib = _lib.lib()
    d_lo, d_hi = model.shard
    outs = torch.zeros((a.steps, 5), dtype=torch.float64, device=dev)   # every step's objective breakdown stays on the device
    host_obj = torch.zeros(a.steps, dtype=torch.float64).pin_memory()   # ... and its objective lands in host memory (async copy)
    # One step = one evaluation from the device-resident raw parameters to the objective IN HOST MEMORY: the scalar is copied
    # to a pinned host buffer on the stream in every step (no host synchronisation per step; one at the end of the region).
    # Launch mode: eager kernel launches, or (--graph on; default when D is sharded) the evaluation replayed from a HIP graph.
    # HIP events around the psi2 kernel on every EV_EVERY-th step only (each recorded event costs ~6 us of stream time; a 20-step
    # run has five samples);
    # those steps are always eager launches.
    use_graph = (a.graph == 'on')          # ('auto' = eager: measured, the graph replay is no faster at D = 64 per GPU and slower at D = 512)
    EV_EVERY = 4
    ev = {i: (lib.dpgp_event_create(), lib.dpgp_event_create()) for i in range(0, a.steps, EV_EVERY)}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        if use_graph and i not in ev:
            model.evaluate_graph(out=outs[i])
        else:
            model.evaluate_(events=ev.get(i), out=outs[i])
        host_obj[i].copy_(outs[i, 0], non_blocking=True)

    if use_graph:
        model.evaluate_graph()
    for _ in range(a.warmup):
        model.evaluate_graph() if use_graph else model.evaluate_()
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
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
    objs = host_obj.numpy().copy()
    terms, info = model.per_dimension_terms
    if os.environ.get('DPGP_BENCH_NOCHECK'):                      # (timing experiments with diagnostic kernel builds only)
        objs[:] = 0.0
    assert np.isfinite(objs).all() and np.all(objs == objs[0]), 'objective is not finite / not reproducible'
    assert os.environ.get('DPGP_BENCH_NOCHECK') or np.array_equal(objs, outs[:, 0].cpu().numpy())
    assert os.environ.get('DPGP_BENCH_NOCHECK') or int(info.abs().max().item()) == 0, 'a factorisation failed or a precision guard fired'

    def rate(mdl, steps, warm=3):
        """evals/s of another model object on the same problem (side figures; same launch mode, short run)."""
        for _ in range(warm):
            mdl.evaluate_()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            mdl.evaluate_()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        assert int(mdl.per_dimension_terms[1].abs().max().item()) == 0
        return steps / dt, float(mdl.objective)

    if rank == 0:
        d_loc = d_hi - d_lo
        # algorithmic work of the dominant kernel (psi2) per launch: SURVEY.md 8(d): per output dim N*M(M+1)/2 exps and
        # (4Q+2) flops per exponent on the symmetric half; one launch processes d_loc output dims.
        exps = d_loc * n * m * (m + 1) // 2
        flops = exps * (4 * q + 2)
        achieved = flops / (psi2_ms * 1e-3) / 1e12
        traffic, traffic_rec = profiled_traffic(a.config, a.prec, world)
        exp_peak = 256 * 4 * 8 * 2.4e9        # v_exp_f32: 64 lanes / 8 cycles per SIMD, 1024 SIMDs, 2.4 GHz
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
                       'precision': a.prec, 'launch': 'hip graph replay' if use_graph else 'eager',
                       'step': 'raw parameters in HBM -> objective in pinned host memory (async copy per step, one sync per run)'},
            'objective': float(objs[0]), 'git_sha': git_sha(),
            # dominant kernel: psi2 (+ the K_uu task slice that rides in the same dispatch).  `achieved` = ALGORITHMIC flops
            # (N M(M+1)/2 exponents x (4Q+2) flops per output dim) / HIP-event duration; `peak` = dense peak of
            # the matrix pipe the kernel runs on (f16 operands, fp32 accumulate; fp64 MFMA for --prec f64).  The matrix pipe
            # is not what limits it: every exponent costs one v_exp_f32 (8 issue cycles per wave) + one accumulate, see
            # `exp_frac` (vs the v_exp_f32 rate alone) and DESIGN.md section 4.
            # `bound` names the pipe that limits the kernel: for the f16-split kernels that is the vector unit's
            # transcendental issue (one v_exp_f32 = 8 issue cycles per wave-instruction -> 256 CUs x 4 SIMDs x 64 / 8 x 2.4 GHz =
            # 1.966e13 exp/s), so `achieved` / `peak` / `frac` are in exponentials per second; the matrix-pipe figure of the
            # same launch (algorithmic flops / dense f16 peak) is kept under `mfma`.  --prec f64: the fp64 matrix pipe.
            'roofline': ({'bound': 'valu-transcendental', 'achieved': exps / (psi2_ms * 1e-3) / 1e9, 'peak': exp_peak / 1e9,
                          'unit': 'Gexp/s', 'frac': exps / (psi2_ms * 1e-3) / exp_peak,
                          'mfma': {'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                                   'vs_fp32_matrix_peak': achieved / MFMA_F32_PEAK_TFLOPS},
                          'limiter': 'VALU issue: one v_exp_f32 (8 cycles) + one v_add_f32 (4) per exponent; the matrix pipe runs beneath'}
                         if f16_path else
                         {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                          'limiter': 'fp64 issue: v_mfma_f64 and fp64 VALU (table exp2) share the pipe and do not overlap'}),
        }
        res['roofline'].update({'traffic': traffic, 'traffic_source': traffic_rec, 'kernel_ms': psi2_ms,
                                'kernel': 'psi2_pairs_kernel (+ K_uu task slice)' if f16_path else 'psi2_mfma_kernel (+ K_uu task slice)',
                                'algorithmic_exps': exps, 'algorithmic_flops': flops})
        if world == 1 and not a.no_secondary:
            res['secondary'] = secondary(dev, shape, p)
        if world == 1 and not a.no_side and a.prec == 'mixed':
            # the same metric in the other arithmetic: exact-fp32 products (v_mfma_f32_16x16x4_f32, no f16 split / range limit)
            # and the reference's own fp64 (src/utils/types.py:13-14) — short runs, same problem, same step definition
            ks = max(3, min(10, a.steps))
            v32, o32 = rate(dp_gp_lvm(p['y'], precision='mixed', psi_algo='mfma_f32', **kw), ks)
            m64 = dp_gp_lvm(p['y'], precision='f64', backward_precision='mixed', **kw)
            v64, o64 = rate(m64, ks)
            res['value_fp32_exact'] = v32
            res['value_f64'] = v64
            # `value` is measured with precision='mixed' (bench.py's --prec default); a model built WITHOUT a precision argument
            # runs the reference's arithmetic: fp64 forward (and fp64 dense adjoints + the matrix-pipe stage B in its gradients)
            res['default_constructed_model'] = {'precision': 'f64', 'backward_precision': 'mixed', 'value': v64,
                                                'unit': 'ELBO evals/s', 'note': 'dp_gp_lvm(...) with no precision argument; '
                                                'the headline `value` needs precision="mixed"'}
            res['precision_check'] = {'objective_mixed': float(objs[0]), 'objective_fp32_exact': o32, 'objective_f64': o64,
                                      'rel_err_mixed_vs_f64': abs(float(objs[0]) - o64) / abs(o64),
                                      'rel_err_fp32_exact_vs_f64': abs(o32 - o64) / abs(o64)}
        if world == 1 and not a.no_grad and (a.prec == 'mixed' or (a.prec == 'f64' and m <= 128)):
            # side measurements, not the headline metric: (1) one objective evaluation + the gradients of all raw variables
            # (backward pass; what one Adam iteration of the reference needs); (2) one objective evaluation
            # of the over-T formulation dp_gp_lvm_t on the same data (8f row 3: T Psi2's instead of D)
            def grad_ms(mdl, reps):
                for _ in range(2):
                    mdl.gradients()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    mdl.gradients()
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - t1) / reps
            reps = 10
            res['objective_and_gradients'] = {'ms': grad_ms(model, reps), 'reps': reps,
                                              'note': 'one training step: dpgp_elbo_step where it applies (mixed, M <= 128, Q <= 20: Psi2 out '
                                                      'of the first pass of stage B, stage A, stage B incl. the Psi1 term through the same '
                                                      'pass kernel) + dpgp_model_backward; otherwise forward + stage A + stage B as separate '
                                                      'calls (patch form of the Psi2 term for Q > 20 and behind an fp64 forward); plain kernel '
                                                      'in f64; DESIGN.md 7.1'}
            # breakdown of one more iteration THROUGH THE SEPARATE CALLS (torch events on the launch stream): forward, stage A,
            # stage B, chain rule — the one-call step above has no stage boundaries to put events on and saves the forward's psi2
            # dispatch (so ms < the sum below); and the exponential rate of stage B over its two passes
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            model.gradients(events=evs[:4])                      # (warm-up of the separate-call path)
            torch.cuda.synchronize()
            evs[4].record()
            model.gradients(events=evs[:4])
            torch.cuda.synchronize()
            t_fwd, t_a, t_b, t_c = (evs[4].elapsed_time(evs[0]), evs[0].elapsed_time(evs[1]), evs[1].elapsed_time(evs[2]),
                                    evs[2].elapsed_time(evs[3]))
            # exponentials of the Psi2 term: two passes over the M (M + 1) / 2 pairs (pair-tile form), or the full square of
            # 64 x 64 patches (patch form)
            mp64 = 64 * ((m + 63) // 64)
            pair_form = a.prec == 'mixed' and q <= 20
            exps_b = float(n) * (d_hi - d_lo) * (2.0 * (m * (m + 1) // 2) if pair_form else mp64 * mp64)
            res['objective_and_gradients'].update({
                'separate_calls': {'forward_ms': t_fwd, 'stage_a_ms': t_a, 'stage_b_ms': t_b, 'chain_rule_ms': t_c},
                'stage_b_exp_per_s': exps_b / (t_b * 1e-3), 'stage_b_exp_frac_of_v_exp_rate': exps_b / (t_b * 1e-3) / exp_peak})
            if a.prec == 'mixed' and q <= 20:
                # opt-in DPGP_PREC_MIXED_FAST for stage B (include/dpgp.h: 11-bit exponentials in its second products; measured <= 2e-5
                # of the largest gradient entry against fp64 at the BASELINE shapes, tests/test_gpu_grad.py): never the default
                res['objective_and_gradients']['fast_stage_b_ms'] = grad_ms(
                    dp_gp_lvm(p['y'], precision='mixed', backward_precision='mixed_fast', **kw), reps)
            if a.prec == 'mixed' and m <= 128:
                # the training configuration that follows the reference's fp64 arithmetic through an Adam run (DESIGN.md
                # section 5): fp64 forward and dense adjoints, streaming stage B on the matrix pipe
                res['objective_and_gradients']['training_configuration_f64_forward_mixed_stage_b_ms'] = grad_ms(
                    dp_gp_lvm(p['y'], precision='f64', backward_precision='mixed', **kw), 5)
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
                                           'note': 'dp_gp_lvm_t objective: dpgp_model_prepare_t + dpgp_elbo_fhat_t (fused, eleven launches on two streams) for M <= 128; '
                                                   'gradients: dpgp_model_prepare_t + the library operators (f_hat and its adjoints, stage B) + '
                                                   'dpgp_model_backward_t, replayed from a HIP graph in optimise()'}
                if a.prec == 'mixed':
                    res['objective_over_t']['with_gradients_ms'] = grad_ms(model_t, reps)
                    # one optimise() iteration of both models (gradients, collective flag, Adam update; the reference's only
                    # performance assertion is that the over-T model trains faster: test/unittests/dpgplvm_unitttests.py:576)
                    def opt_ms(mdl):
                        mdl.optimise(3)
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        mdl.optimise(reps)
                        torch.cuda.synchronize()
                        return 1e3 * (time.perf_counter() - t1) / reps
                    res['objective_over_t']['optimise_iteration_ms'] = opt_ms(model_t)
                    res['objective_and_gradients']['optimise_iteration_ms'] = opt_ms(model)
        if not a.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline(a.config, p, shape, a.cpu_dims)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
