"""Cycle stamps of the single-launch index build (a library whose tmx_index.hip was built with -DTMX_INDEX_STAMPS, given by
TMX_LIB_PATH), one 32k prompt, the L2 flushed before every launch as a 170 us attention launch leaves it."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops, _lib
dev = torch.device("cuda:0")
n = 32768
cu_h, sl_h = np.asarray([0, n], np.int32), np.asarray([n], np.int32)
cu, sl = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
bt = torch.randperm(n // 16).to(torch.int32).view(1, -1).to(dev)
lib = _lib.load()
idx = ops.build_block_index(cu, sl, cu_h, sl_h, 4, 32, 64, 32, 32, 2, chunk_tiles=64, block_table=bt, page_size=16)
big = torch.empty(64 << 20, dtype=torch.float32, device=dev)
res = []
for it in range(12):
    big.fill_(1.0)  # push the batch description and the block table out of the L2, as a 170 us attention launch does
    _lib.check(lib.tmx_build_block_index(ctypes.byref(idx.c), cu.data_ptr(), sl.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 16)()
    lib.tmx_index_read_stamps(buf)
    a = np.array(buf[:12], dtype=np.int64)
    res.append([a[1] - a[0], a[7] - a[1], a[8] - a[7], a[9] - a[8], a[3] - a[9], a[5] - a[3], a[6] - a[5], a[10] - a[9], a[11] - a[10], a[3] - a[11]])
r = np.median(np.array(res[2:]), axis=0)
print("cycles: batch+scan1 %d | counts %d | scan2 %d | stores %d | short fill+sync %d | long lists %d | max %d   (short fill: lists %d, long-list table %d, barrier %d)" % tuple(r))
