"""dev: why is the host side of a step sometimes 2x slower (bench 416 instead of 525 scenes/s, same kernels)?  For this
process: the CPU it runs on, how busy that CPU's SMT sibling and the whole host are (from /proc/stat), the host time of
4000 tiny launches there -- and again after moving to the idlest core of the GPU's NUMA node."""
import os
import sys
import time

import torch


def cpu_now():
    return int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])  # field 39: the CPU last run on


def stat():
    out = {}
    for line in open("/proc/stat"):
        if line.startswith("cpu") and line[3].isdigit():
            f = line.split()
            v = list(map(int, f[1:9]))
            out[int(f[0][3:])] = (sum(v), v[3] + v[4])  # total, idle + iowait
    return out


def busy(a, b):
    return {c: 1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in a}


def sibling(c):
    try:
        lst = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        return [int(x) for x in lst.replace("-", ",").split(",") if int(x) != c]
    except OSError:
        return []


def measure(x, n=4000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        x.add_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e6 * (t1 - t0) / n


x = torch.zeros(64, device="cuda")
measure(x, 200)
s0 = stat()
time.sleep(0.2)
s1 = stat()
load = busy(s0, s1)
c = cpu_now()
us = measure(x)
tot = sum(load.values())
print(f"cpu {c}: {us:.2f} us per launch; before: cpu {c} busy {load[c]:.2f}, sibling {sibling(c)} busy "
      f"{[round(load[s], 2) for s in sibling(c)]}; host: {tot:.1f} CPUs busy of {len(load)}, "
      f"busiest {sorted(((round(v, 2), k) for k, v in load.items()), reverse=True)[:6]}")
if len(sys.argv) > 1 and sys.argv[1] == "move":
    p = torch.cuda.get_device_properties(0)
    bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    cpus = set()
    for part in open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    best = min(cpus, key=lambda k: load[k] + sum(load[s] for s in sibling(k)))
    os.sched_setaffinity(0, {best})
    print(f"   moved to cpu {best} (busy {load[best]:.2f}, sibling {[round(load[s], 2) for s in sibling(best)]}): "
          f"{measure(x):.2f} us per launch")
