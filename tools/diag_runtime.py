import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l})
if order == 'torch_first':
    import torch; torch.zeros(1).cuda()
    from cilantro_amd import capi; L = capi.load()
else:
    from cilantro_amd import capi; L = capi.load()
    import torch; torch.zeros(1).cuda()
h = ctypes.c_void_p()
print(order, 'create rc =', L.cilhip_create(ctypes.byref(h), 0))
print('\n'.join(maps()))
