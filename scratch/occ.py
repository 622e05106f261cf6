import ctypes, os, torch
lib = ctypes.CDLL(os.environ['DPGP_LIBRARY'])
torch.cuda.init()
for extra in (0, -2048, -4096, -8192, -16384):
    print('chain_b occupancy (blocks/CU * 1e6 + lds bytes), extra', extra, lib.dpgp_debug_chain_b_occupancy(128, extra))
