#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
python - > $O/x2_parity.txt 2>&1 <<'PY'
import ctypes, torch, numpy as np, sys
sys.path.insert(0, '.')
from palettenerf_amd import _lib, gridencoder
from palettenerf_amd._torch_glue import call, ptr
lib = _lib.load()
dev = torch.device('cuda', 0)
enc = gridencoder.GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=4096).to(dev)
enc.embeddings.data.uniform_(-0.5, 0.5)
for B in (1, 255, 513, 70001, 626627):
    x = torch.rand(B, 3, device=dev); 
    if B > 10: x[3] = 1.5; x[7, 1] = -0.2; x[5] = 1.0; x[6] = 0.0
    outs = []
    for nt in (0, 4):
        lib.pnr_set_option(b"grid_nt", nt)
        out = torch.full((16 * B * 2,), 7.0, device=dev)
        call("pnr_grid_encode_forward_layout", ptr(x), ptr(enc.embeddings.detach()), ptr(enc.offsets), ptr(out), ctypes.c_uint32(B), ctypes.c_uint32(3), ctypes.c_uint32(2), ctypes.c_uint32(16),
             ctypes.c_float(np.log2(enc.per_level_scale)), ctypes.c_uint32(16), None, ctypes.c_uint32(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0))
        outs.append(out.clone())
    lib.pnr_set_option(b"grid_nt", 0)
    print(B, "bit-identical:", bool(torch.equal(outs[0], outs[1])))
PY
timeout 600 python profiles/grid_op_bench.py > $O/grid_op_bench.txt 2>&1
