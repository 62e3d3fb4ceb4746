"""One process, several devices: the C entry of the sharded ICP loop (include/cilantro_hip/c_api.h, cilhip_multi_*).

The sharded protocols of cilantro_amd/distributed.py (one process per GPU over torch.distributed) exist a second time below the
C ABI, for callers without Python / torch: one context per device, RCCL's all-reduce of the 48 partial sums on the devices'
streams, the slab guard and the re-partitioning handled inside the library; PARTITION_TARGET_SHARDS: the target in index shards, a MIN
all-reduce of one packed (d2, global index) key per source point and iteration, the reference's tie order across shards through a
second key (c_api.h: cilhip_icp_order_keys).  This module is the thin ctypes mirror of
``cilantro_hip::MultiDeviceRigidICP`` (include/cilantro_hip/icp.hpp).
"""
import ctypes as C

import numpy as np

from . import capi

PARTITION_SOURCE_SHARDS, PARTITION_SLABS, PARTITION_TARGET_SHARDS = 0, 1, 2


class MultiDeviceRigidICP:
    def __init__(self, devices):
        self._L = capi.load()
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = self._L.cilhip_multi_create(C.byref(self._h), arr, len(devices))
        if rc != 0:
            raise RuntimeError(f"cilhip_multi_create failed ({rc}): no usable HIP device / RCCL for these ordinals")
        self.n = len(devices)

    def close(self):
        if self._h:
            self._L.cilhip_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(self._L.cilhip_multi_last_error(self._h).decode())

    def set_clouds(self, dst, dst_normals, src, max_sq_dist, partition=PARTITION_SLABS, T_part=None):
        dst = np.ascontiguousarray(dst, np.float32).reshape(-1, 3); src = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
        dn = None if dst_normals is None else np.ascontiguousarray(dst_normals, np.float32).reshape(-1, 3)
        tp = None if T_part is None else np.ascontiguousarray(np.asarray(T_part, np.float32).reshape(4, 4).T).reshape(16)
        self._ck(self._L.cilhip_multi_set_clouds(self._h, dst.ctypes.data, dn.ctypes.data if dn is not None else None, len(dst), src.ctypes.data, len(src),
                                                 float(max_sq_dist), int(partition), tp.ctypes.data if tp is not None else None))

    def set_option(self, key, value):
        for r in range(self.n):
            ctx = self._L.cilhip_multi_context(self._h, r)
            if self._L.cilhip_set_option(ctx, key.encode(), float(value)) != 0:
                raise RuntimeError(self._L.cilhip_last_error(ctx).decode())

    def icp_run(self, params, T0=None, check_every=0):
        res = capi.IcpResult()
        t0 = None if T0 is None else np.ascontiguousarray(np.asarray(T0, np.float32).reshape(4, 4).T).reshape(16)
        self._ck(self._L.cilhip_multi_icp_run(self._h, C.byref(params), t0.ctypes.data if t0 is not None else None, int(check_every), C.byref(res)))
        return res

    def set_slab_slack(self, slack):
        """how far a source point may move along the slab axis before the slabs are cut again (< 0: twice the search radius)"""
        self._ck(self._L.cilhip_multi_set_slab_slack(self._h, float(slack)))

    def last_host_time(self):
        """host microseconds per iteration the (slowest) shard's enqueue calls took in the last icp_run"""
        v = C.c_double(0.0)
        self._L.cilhip_multi_last_host_time.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._ck(self._L.cilhip_multi_last_host_time(self._h, C.byref(v)))
        return v.value

    def repartitions(self):
        return int(self._L.cilhip_multi_repartitions(self._h))

    def shard_sizes(self, rank):
        a = C.c_size_t(0); b = C.c_size_t(0)
        self._ck(self._L.cilhip_multi_shard_sizes(self._h, rank, C.byref(a), C.byref(b)))
        return a.value, b.value
