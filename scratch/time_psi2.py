# scratch: time the standalone psi2 launch (no K_uu task slice) at a bench configuration; DPGP_LIBRARY selects a variant build
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = make_problem(cfg)
dev = torch.device('cuda', 0)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
z, mu, s, g, al = t(p['z']), t(p['mu']), t(p['s']), t(p['gamma']), t(p['alpha'])
for _ in range(3): out = ops.psi2(z, mu, s, g, al)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out = ops.psi2(z, mu, s, g, al)
e1.record(); torch.cuda.synchronize()
print('%s cfg %d psi2 (+finish) %.4f ms  checksum %.6e' % (os.environ.get('DPGP_LIBRARY', 'default'), cfg, e0.elapsed_time(e1) / 20, float(out.double().sum())))
if 'prof' in os.environ.get('DPGP_LIBRARY', ''):
    import ctypes
    st = (ctypes.c_longlong * 16)()
    lib = ctypes.CDLL(os.environ['DPGP_LIBRARY'])
    lib.dpgp_debug_psi2_stamps(st)
    u = lambda x: x / 100.0
    print('   one workgroup, wave 0 (us): prologue %.1f  phaseA %.1f  phaseB %.1f  phaseC %.1f  loop total %.1f  epilogue %.1f  total %.1f' %
          (u(st[1] - st[0]), u(st[2]), u(st[3]), u(st[4]), u(st[5] - st[1]), u(st[6] - st[5]), u(st[6] - st[0])))
    if hasattr(lib, 'dpgp_debug_psi2_wg'):
        n_, d_, m_, q_ = p['y'].shape[0], p['gamma'].shape[0], p['z'].shape[0], p['z'].shape[1]
        nwg = min(8192, d_ * 3 * 8)
        buf = (ctypes.c_longlong * (3 * nwg))()
        lib.dpgp_debug_psi2_wg(buf, nwg)
        a = np.array(buf[:], dtype=np.int64).reshape(nwg, 3)
        a = a[a[:, 1] > 0]
        t0 = a[:, 0].min()
        st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
        print('   %d workgroups: makespan %.1f us; duration mean %.1f min %.1f max %.1f us; sum/makespan = %.1f concurrent' %
              (len(a), en.max(), (en - st).mean(), (en - st).min(), (en - st).max(), (en - st).sum() / en.max()))
        xcc = (a[:, 2] >> 32) & 0xf
        hw = a[:, 2] & 0xffffffff
        cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
        slot = xcc * 1000 + se * 100 + sh * 16 + cu
        print('   distinct (xcc,se,sh,cu):', len(np.unique(slot)), ' WGs per xcc:', np.bincount(xcc.astype(int)))
        for lo in range(0, int(en.max()) + 1, max(1, int(en.max()) // 12)):
            print('   t=%5d us: running %d' % (lo, int(((st <= lo) & (en > lo)).sum())))
        third = len(a) // 3
        for k in range(3):
            sl = slice(k * third, (k + 1) * third)
            print('   grid z=%d: start %.0f..%.0f  duration mean %.1f' % (k, st[sl].min(), st[sl].max(), (en[sl] - st[sl]).mean()))
        d0 = (en - st)[:third]
        print('   z=0 duration by xcc:', ' '.join('%.0f' % d0[xcc[:third] == x].mean() for x in range(8)))
        print('   z=0 duration by se :', ' '.join('%.0f' % d0[se[:third] == x].mean() for x in np.unique(se)))
        print('   z=0 duration by cu :', ' '.join('%.0f' % d0[cu[:third] == x].mean() for x in np.unique(cu)))
        print('   z=0 duration by sh :', ' '.join('%.0f' % d0[sh[:third] == x].mean() for x in np.unique(sh)))
        s0 = slot[:third]
        pairs = [d0[s0 == u] for u in np.unique(s0)]
        print('   z=0 WGs per CU:', np.bincount([len(x) for x in pairs]), ' mean |diff| within CU %.1f' % np.mean([abs(x[0] - x[1]) for x in pairs if len(x) == 2]))
        print('   z=0 duration percentiles:', np.percentile(d0, [0, 10, 25, 50, 75, 90, 100]).round(0))
        simd = (hw >> 4) & 3; wave = hw & 0xf
        print('   wave ids', np.bincount(wave.astype(int)), 'simd ids', np.bincount(simd.astype(int)))
