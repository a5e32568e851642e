#!/usr/bin/env python3
"""Dev tool (GPU box): the memory controllers' activity beside a running command -- the HBM side of the roofline that no
rocprofv3 counter shows (profiles/r06_dram_calib.md: every memory-side TCC counter, TCC_EA0_RDREQ_DRAM* included, counts requests
that the Infinity Cache serves).

    umc_sample.py [--period-ms 20] [--out file.json] -- command ...

Samples /sys/class/drm/card*/device/mem_busy_percent (the SMU's average UMC activity, what `rocm-smi --showmemuse` prints as "GPU
Memory Read/Write Activity") and gpu_busy_percent of every card while the command runs, and reports, per card that was busy, the
mean / median / maximum over the middle 80 % of the run.  The figure is a firmware average in whole percent: calibrate it on
known patterns first (tools/ubench/dram_calib loop read|reread|..., tools/archive/r06_b.sh) before reading a workload with it."""
import glob, json, os, statistics, subprocess, sys, time

args = sys.argv[1:]
period, out = 0.02, None
while args and args[0] != "--":
    if args[0] == "--period-ms":
        period = float(args[1]) / 1e3; args = args[2:]
    elif args[0] == "--out":
        out = args[1]; args = args[2:]
    else:
        sys.exit(__doc__)
cmd = args[1:]
if not cmd:
    sys.exit(__doc__)
cards = sorted(glob.glob("/sys/class/drm/card*/device/mem_busy_percent"))
# which card is HIP device 0 of this box (a GPU box sees one GPU of a host whose sysfs lists all of them): by PCI address, asked in
# a short-lived child so that this process never holds the device
mine = None
try:
    code = ("import ctypes; h = ctypes.CDLL('libamdhip64.so'); b = ctypes.create_string_buffer(64); "
            "print(b.value.decode() if h.hipDeviceGetPCIBusId(b, 64, 0) == 0 and b.value else '')")
    bus = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60).stdout.strip().lower()
    for c in cards:
        if bus and os.path.basename(os.path.realpath(os.path.dirname(c))).lower() == bus:
            mine = c
    if mine:
        cards = [mine]
except Exception:
    pass


def read(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return None


def accumulators():
    """(timestamp, gfx_activity_acc, mem_activity_acc) of rocm_smi device 0 -- the accumulating twins of the two percentages
    (`rocm-smi --showmemuse`: "Memory Activity"), or None."""
    import ctypes
    try:
        L = ctypes.CDLL("librocm_smi64.so")
        if not getattr(accumulators, "up", False):
            if L.rsmi_init(ctypes.c_uint64(0)) != 0:
                return None
            accumulators.up = True

        class Ctr(ctypes.Structure):
            _fields_ = [("type", ctypes.c_int), ("val", ctypes.c_uint64)]
        arr = (Ctr * 2)()
        arr[0].type, arr[1].type = 0, 1
        ts = ctypes.c_uint64(0)
        if L.rsmi_utilization_count_get(ctypes.c_uint32(0), arr, ctypes.c_uint32(2), ctypes.byref(ts)) != 0:
            return None
        return ts.value, arr[0].val, arr[1].val
    except OSError:
        return None


acc0 = accumulators()
t0 = time.time()
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
samples = []
while p.poll() is None:
    row = [time.time() - t0]
    for c in cards:
        row.append(read(c)); row.append(read(c.replace("mem_busy_percent", "gpu_busy_percent")))
    samples.append(row)
    time.sleep(period)
so, se = p.communicate()
wall = time.time() - t0
acc1 = accumulators()
lo, hi = int(len(samples) * 0.1), max(int(len(samples) * 0.9), 1)
mid = samples[lo:hi] or samples
rec = {"card_of_hip_device_0": mine.split("/")[4] if mine else None, "command": " ".join(cmd)[:300], "wall_s": round(wall, 3), "samples": len(samples), "period_ms": period * 1e3, "returncode": p.returncode, "cards": {}}
for i, c in enumerate(cards):
    mem = [r[1 + 2 * i] for r in mid if r[1 + 2 * i] is not None]
    gpu = [r[2 + 2 * i] for r in mid if r[2 + 2 * i] is not None]
    if mem and (max(mem) > 0 or (gpu and max(gpu) > 0)):
        rec["cards"][c.split("/")[4]] = {"mem_busy_mean": round(statistics.mean(mem), 2), "mem_busy_median": statistics.median(mem), "mem_busy_max": max(mem),
                                         "gpu_busy_mean": round(statistics.mean(gpu), 2) if gpu else None}
if acc0 and acc1:
    rec["accumulators"] = {"timestamp_delta": acc1[0] - acc0[0], "gfx_activity_delta": acc1[1] - acc0[1], "mem_activity_delta": acc1[2] - acc0[2],
                           "mem_activity_per_s": round((acc1[2] - acc0[2]) / wall, 1)}
last = [l for l in so.strip().split("\n") if l.startswith("{")]
if last:
    try:
        rec["stdout_json"] = json.loads(last[-1])
    except ValueError:
        pass
if p.returncode != 0:
    rec["stderr_tail"] = se[-400:]
txt = json.dumps(rec)
print(txt)
if out:
    with open(out, "a") as f:
        f.write(txt + "\n")
