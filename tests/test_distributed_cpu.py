"""
CPU test of the multi-GPU composition (world size 2, gloo): the per-output-dimension terms shard over D with one packed
sum all-reduce of (f_hat share, DP-objective share); rank 0 alone adds the D-independent DP terms.  The per-shard
numbers come from the CPU oracle here (the HIP kernels need a GPU); what is under test is the decomposition the model
object uses: shard_bounds, the (f_hat, dp) packing, the all-reduce and the final combination
objective = dp - (f_hat - KL) - hyper-prior   (reference dp_gp_lvm.py:154).
"""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dp_gp_lvm_amd.models.dp_gp_lvm import shard_bounds
from dp_gp_lvm_amd.utils.synthetic import make_problem
from oracle import dpgp_oracle as orc


def _dp_share(phi_rows, g1, g2, w1, w2, s1, s2, add_constants):
    """This rank's share of the DP objective, as dpgp_model_prepare defines it: the d-dependent terms for its rows, plus
    (rank 0 only) everything that does not depend on d."""
    full = orc.dp_objective(phi_rows, g1, g2, w1, w2, s1, s2)
    # d-independent part = objective of an "empty" row set; obtain it by linearity from a 1-row evaluation
    one = orc.dp_objective(phi_rows[:1], g1, g2, w1, w2, s1, s2)
    two = orc.dp_objective(np.concatenate([phi_rows[:1], phi_rows[:1]]), g1, g2, w1, w2, s1, s2)
    const = 2.0 * one - two
    return full if add_constants else full - const


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = make_problem(1)
    d = p['y'].shape[1]
    lo, hi = shard_bounds(d, rank, world)
    terms = orc.fhat_terms(p['y'][:, lo:hi], p['z'], p['mu'], p['s'], p['gamma'][lo:hi], p['alpha'][lo:hi],
                           p['beta'][lo:hi])
    red = torch.tensor([terms.sum(), _dp_share(p['phi'][lo:hi], p['g1'], p['g2'], p['w1'], p['w2'], p['s1'], p['s2'],
                                               rank == 0)], dtype=torch.float64)
    dist.all_reduce(red, op=dist.ReduceOp.SUM)
    kl = orc.kl_qx(p['mu'], p['s'])
    hyper = orc.hyperprior(p['gamma_atoms'], p['alpha_atoms'], p['beta_atoms'])
    obj = float(red[1]) - (float(red[0]) - kl) - hyper
    if rank == 0:
        out.put((obj, float(red[0]), float(red[1])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_objective_matches_single_process():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    obj, fhat, dpo = out.get(timeout=120)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    p = make_problem(1)
    ref = orc.objective(p['y'], p['z'], p['mu'], p['s'], p['phi'], p['gamma_atoms'], p['alpha_atoms'], p['beta_atoms'],
                        p['g1'], p['g2'], p['w1'], p['w2'], p['s1'], p['s2'])
    np.testing.assert_allclose(obj, ref, rtol=1e-12)
    np.testing.assert_allclose(dpo, orc.dp_objective(p['phi'], p['g1'], p['g2'], p['w1'], p['w2'], p['s1'], p['s2']),
                               rtol=1e-11)
