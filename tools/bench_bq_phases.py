#!/usr/bin/env python3
"""Dev: time the tile ball query's phases (CODA_BQ_TILE_STOP=1..3 returns after phase A / B / C; 0 = all)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from coda_neurips2023_amd.pointnet2 import _ext
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    dev = torch.device("cuda:0")
    pc, _, _ = make_batch(8, 20000, seed=1234)
    xyz = torch.from_numpy(pc).to(dev)
    inds = _ext.furthest_point_sampling(xyz, 2048)
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    for _ in range(3):
        _ext.query_and_group_xyz(new_xyz, xyz, 0.2, 64, True, algorithm="tile", channels_last=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); _ext.query_and_group_xyz(new_xyz, xyz, 0.2, 64, True, algorithm="tile", channels_last=True); e.record()
        torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    print(f"stop_after={os.environ.get('CODA_BQ_TILE_STOP','0')}: median {np.median(ts)*1e3:.1f} us min {np.min(ts)*1e3:.1f} us")
else:
    for st in ("1", "2", "3", "0"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, CODA_BQ_TILE_STOP=st))
