"""Reads gpurun_out/step_trace.json.gz (tools/op_attribution.py) and attributes every PyTorch-native kernel of one
training step to the innermost frame of this package that launched it (backward nodes: the autograd node)."""
import collections
import gzip
import json
import sys

d = json.load(gzip.open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_trace.json.gz"))
ev = d["traceEvents"]
py = [e for e in ev if e.get("cat") == "python_function" and "dur" in e]
ops = [e for e in ev if e.get("cat") == "cpu_op" and "dur" in e]
kern = [e for e in ev if e.get("cat") in ("kernel", "gpu_memset", "gpu_memcpy")]
corr_rt = {e["args"].get("correlation"): e for e in ev if e.get("cat") == "cuda_runtime" and "args" in e}
by_tid, ops_by_tid = collections.defaultdict(list), collections.defaultdict(list)
for e in py:
    by_tid[e["tid"]].append(e)
for e in ops:
    ops_by_tid[e["tid"]].append(e)
for t in ops_by_tid:
    ops_by_tid[t].sort(key=lambda e: (e["ts"], -e["dur"]))
agg = collections.defaultdict(lambda: [0.0, 0])
for k in kern:
    r = corr_rt.get(k["args"].get("correlation"))
    if r is None:
        continue
    kn = k["name"]
    if not (kn.startswith("void at::") or "rocclr" in kn or k.get("cat") != "kernel" or "at::native" in kn):
        continue
    ts, tid = r["ts"], r["tid"]
    frames = [f["name"] for f in by_tid.get(tid, []) if f["ts"] <= ts < f["ts"] + f["dur"]
              and ("coda_neurips2023_amd" in f["name"] or "bench.py" in f["name"])]
    opsl = [e for e in ops_by_tid.get(tid, []) if e["ts"] <= ts < e["ts"] + e["dur"]]
    where = frames[-1].split("coda_neurips2023_amd/")[-1] if frames else "bwd:" + (opsl[0]["name"] if opsl else "?")
    key = (opsl[-1]["name"] if opsl else "?", where)
    agg[key][0] += k["dur"]
    agg[key][1] += 1
print("torch-native kernels: %.1f us/step, %d launches" % (sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())))
for (op, where), (dt, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 50]:
    print(f"{dt:8.1f} us n={n:3d} {op[:28]:28s} {where[:110]}")
