"""Child process of tests/test_gpu_sharded.py: one rank of a D-sharded dp_gp_lvm on the visible GPU.
    python tests/_sharded_worker.py <fixture> <precision> <out.npz> <mode>
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the environment.  The process group is gloo (its collectives take
device tensors): RCCL refuses two ranks on one device, and the GPU box of the test run has one.  What is under test is the
product's own sharded path — shard_bounds, dpgp_model_prepare with d_offset / add_constants, pack -> all_reduce ->
dpgp_model_finalize, the packed gradient all-reduce and the collective trouble flag of optimise() — which is independent of
the transport."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests')]


def main():
    fixture, prec, out, mode = sys.argv[1:5]
    import torch
    import torch.distributed as dist
    from conftest import golden
    from test_gpu_grad import build_model
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda:0')
    g = golden(fixture)
    res = dict(rank=rank)
    if mode == 'values':
        model = build_model(g, dev, prec, process_group=dist.group.WORLD)
        res['objective'] = float(model.objective)
        res['terms'] = model.objective_terms.cpu().numpy()
        res['shard'] = np.array(model.shard)
        for k, v in model.gradients().items():
            res['grad_' + k] = v.cpu().numpy()
        before = res['objective']
        stats = model.optimise(5, learning_rate=0.01)
        res['after'] = float(model.objective)
        res['descended'] = int(res['after'] < before)
        res['x_u_after'] = model.raw['x_u'].cpu().numpy()
        res['precision'] = stats['precision']
    elif mode == 'values_t':
        # the over-T model (dp_gp_lvm_t), D-sharded: objective terms, all gradients, a 5-step Adam run
        from functools import partial
        from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm_t
        from test_gpu_model_t import build
        model = build(partial(dp_gp_lvm_t, process_group=dist.group.WORLD), g, dev, prec)
        res['shard'] = np.array(model.shard)
        res['terms'] = model.objective_terms.cpu().numpy()
        res['terms_graph'] = model.objective_terms_graph().cpu().numpy()
        for k, v in model.gradients().items():
            res['grad_' + k] = v.cpu().numpy()
        model.optimise(5, learning_rate=0.01)
        res['after'] = float(model.objective)
        res['x_u_after'] = model.raw['x_u'].cpu().numpy()
    elif mode == 'flag':
        # an ill-conditioning flag on ONE rank's output dims only: both ranks must raise in the same iteration (the flag
        # travels with the packed gradients), neither may be left waiting in the next all-reduce
        # output dims of rank 0 -> atom 0 (unchanged), output dims of rank 1 -> atom 1 with ARD weights x 1e-5 (K_uu ~ singular)
        bad = {k: np.array(v) for k, v in g.items()}
        d = bad['y'].shape[1]
        logits = np.full_like(bad['dp_logits'], -20.0)
        logits[: d // 2, 0] = 20.0
        logits[d // 2:, 1] = 20.0
        bad['dp_logits'] = logits
        sp = np.logaddexp(0.0, bad['gamma_atoms_raw'])
        sp[1] *= 1e-5
        bad['gamma_atoms_raw'] = np.log(np.expm1(sp))
        model = build_model(bad, dev, 'mixed', process_group=dist.group.WORLD)
        try:
            model.optimise(3)
            res['raised'] = 0
        except FloatingPointError:
            res['raised'] = 1
        res['local_flags'] = int((model.per_dimension_terms[1] != 0).sum())
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
