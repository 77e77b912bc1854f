#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s of the MI355X Mode S receive path (BASELINE.json metric).

A "step" is one pass of the whole hot path (IQ->magnitude, preamble scan, bit slicing, CRC, scoring,
ordered resolve) over one synthetic 2.4 MSPS capture that is already resident in HBM when the timed
region starts.  At N=1 the workload is BASELINE.json configs[1]: a 1 GiB UC8 capture, Mode S only,
--no-fix.  At N>1 every rank replays its own 1 GiB capture (seed 10901+rank) on its own GPU: the
path shards by independent capture, so there is no data-path collective ("scaling": "weak").

One JSON line is printed by rank 0; see the contract in the task description for the keys.
`roofline` is for the dominant kernel (msd_scan_kernel): algorithmic bytes = 2 B per UC8 sample
 x the samples one launch scans, divided by that kernel's average launch duration
measured with HIP events on the stream it runs on (msd_timing.scan_kernel_ms) for every fourth launch of the
timed region -- a hipEventRecord holds the stream for ~5 us, three per launch are 3 % of the whole-job rate.
`cpu_baseline` is the oracle (our CPU restatement, kind "port") on one host core over a bounded
sample of the same capture; it is a reported baseline, never the thing shipped.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-seconds", type=float, default=10.0,
                    help="untimed passes over the capture before the warm-up steps until their time has settled, at most this long")
    ap.add_argument("--samples", type=int, default=1 << 29, help="samples per capture (1 GiB UC8 = 2^29)")
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU batch (0: a quarter of the capture, between 2^26 and 2^27)")
    ap.add_argument("--format", default="uc8", choices=["uc8", "sc16", "sc16q11"])
    ap.add_argument("--fix", type=int, default=0, help="nfix_crc (0 = --no-fix, the configs[1] setting)")
    ap.add_argument("--msgs-per-sec", type=int, default=2000)
    ap.add_argument("--noise-fs", type=float, default=0.02, help="sigma of the I/Q noise in units of full scale (SURVEY.md 8(d): 0.02)")
    ap.add_argument("--threshold", type=int, default=58, help="--preamble-threshold (readsb.c:503-505: 40..400, default 58)")
    ap.add_argument("--input", default="siggen", choices=["siggen", "random"],
                    help="random: uniformly random bytes instead of the synthetic capture (candidate-density sweeps)")
    ap.add_argument("--cpu-sample", type=int, default=1 << 29, help="samples the CPU baseline replays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true",
                    help="diff the GPU message list against the oracle even without the CPU baseline leg (it is on by "
                         "default whenever the baseline replays the whole capture)")
    ap.add_argument("--no-check", action="store_true", help="skip the message-set diff against the oracle")
    ap.add_argument("--no-dropin", action="store_true", help="skip the replay-tool runs of the literal mag_buf path (about 15 s, 10 of them in real time)")
    ap.add_argument("--mode-ac", action="store_true", help="BASELINE configs[4]: Mode A/C demodulator on, 500 replies/s")
    ap.add_argument("--fields", action="store_true",
                    help="MSD_CFG_DECODE_FIELDS: also decode header and extended squitter fields of every message")
    ap.add_argument("--sc16q11-table-bits", type=int, default=0,
                    help="--format sc16q11 only: the converter of a reference built with -DSC16Q11_TABLE_BITS=n (convert.c:264-328)")
    ap.add_argument("--dcfilter", action="store_true",
                    help="MSD_CFG_DC_FILTER: the DC-blocking converters (sequential by nature, ~0.1 GS/s); use a small --samples")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the post-clock runs of BASELINE configs[2] and [4] that the default N=1 run appends ('also')")
    ap.add_argument("--no-pin", action="store_true", help="leave the process's CPU affinity alone")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process group of the N>1 bookkeeping (barrier, MAX time, SUM counts): nccl = RCCL, one GPU per "
                         "rank; gloo = CPU tensors, which also works with several ranks on one GPU "
                         "(MSD_BENCH_DEVICE_OVERRIDE=<index>: every rank uses that device -- tests/test_gpu_bench_two_ranks.py)")
    ap.add_argument("--timing-interval", type=int, default=3,
                    help="record the kernel timing events on one launch in N (msd_set_timing_interval); 3 is coprime "
                         "to the four batches of a pass, so the timed launch rotates over all of them")
    return ap.parse_args()


def pin_to_gpu_local_cpus(torch, local_rank, world, gpu_of=None):
    """The rank's threads (this one, the context's helper thread, the host resolver's pool) onto CPUs of the NUMA
    node its GPU hangs off, a disjoint slice per rank -- the reference pins its reader and demodulator threads too
    (readsb.c:275,749).  Every rank runs a calling thread that polls for events and a helper thread that copies the
    records: eight ranks on one node must not end up on each other's cores.  gpu_of(rank) = the device
    index a local rank uses (default: its own).  Returns a description, or None where sysfs does not tell (nothing is
    changed then)."""
    gpu_of = gpu_of or (lambda r: r)
    try:
        def cpulist(i):
            p = torch.cuda.get_device_properties(i)
            bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            txt = open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip()
            cpus = []
            for part in txt.split(","):
                a, _, b = part.partition("-")
                cpus.extend(range(int(a), int(b or a) + 1))
            return bdf, cpus
        bdf, cpus = cpulist(gpu_of(local_rank))
        allowed = sorted(set(cpus) & os.sched_getaffinity(0))
        if not allowed:
            return None
        mates = [r for r in range(max(world, 1)) if cpulist(gpu_of(r))[1] == cpus]  # local ranks whose GPU shares these CPUs
        k, j = max(1, len(mates)), (mates.index(local_rank) if local_rank in mates else 0)
        per = max(2, len(allowed) // k)
        mine = allowed[j * per:(j + 1) * per] or allowed
        os.sched_setaffinity(0, mine)
        return {"gpu": bdf, "cpus": "%d-%d (%d of the %d local to the GPU, slice %d of %d)" % (mine[0], mine[-1], len(mine), len(allowed), j, k),
                "cpu_list": mine}
    except Exception:  # noqa: BLE001 -- placement is an optimisation: whatever sysfs or the device query says, the run goes on
        return None


def run_also(extra):
    """One of the other BASELINE workloads through this same script in a process of its own, after the clock stopped:
    its bench line, cut down to what the `also` block reports."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--check", "--no-also"] + extra
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        d = json.loads(res.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline line
        return {"workload": " ".join(extra), "error": repr(e)[:200]}
    pm = d.get("pipeline_ms") or {}
    scan, rest = pm.get("scan_kernel_ms"), pm.get("other_kernels_ms")
    return {"workload": d["config"]["workload"], "options": " ".join(extra), "value": d["value"], "unit": d["unit"],
            "ms_per_step": d["ms_per_step"], "dominant_kernel": "msd_scan_kernel" if (scan or 0) >= (rest or 0) else "kernels behind the scan",
            "kernel_ms": scan, "kernels_behind_the_scan_ms": rest, "frac": d["roofline"]["frac"],
            "traffic": d["roofline"]["traffic"], "traffic_of_kernels_behind_the_scan": d["roofline"].get("traffic_of_kernels_behind_the_scan"),
            "messages_per_step": d["messages_per_step"],
            "message_set_diff_vs_oracle": d.get("message_set_diff_vs_oracle"),
            "message_set_diff_vs_second_reading": (d.get("message_set_diff_vs_second_reading") or {}).get("diff"),
            "resolve_stage": d.get("resolve_stage")}


COUNTERS = ("demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted",
            "demod_preamblePhase", "demod_bestPhase", "demod_modeac")


def diff_against_oracle(got, gstats, want, wstats):
    """Entries of the ordered message list (every field the demodulator determines) and demodulator counters that differ."""
    ndiff = abs(len(got) - len(want))
    m = min(len(got), len(want))
    differs = np.zeros(m, dtype=bool)
    for f in ("timestampMsg", "sysTimestampMsg", "signalLevel", "addr", "msgtype", "correctedbits", "score", "crc",
              "bestphase"):
        differs |= got[f][:m] != want[f][:m]
    differs |= (got["msg"][:m] != want["msg"][:m]).any(axis=1)
    return ndiff + int(differs.sum()) + sum(1 for k in COUNTERS if gstats[k] != wstats[k])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def two_thread_stream(O, pkg, iq, bps, ofmt, args, nb2, cpus=None, result=None):
    """The reference's own two-thread structure for one stream (readsb.c:271-285 reader thread: read + convert into
    mag_bufs; readsb.c:820-855 main thread: demodulate), buffers handed over through a 12-deep queue like fifo.c's,
    thread CPU time taken with CLOCK_THREAD_CPUTIME_ID as util.c:102-115 does.  cpus = (reader cpu, demodulator cpu):
    each thread pins itself like readsb.c:275,749.  Runs on the calling thread (the demodulator) + one more."""
    import queue
    import threading
    q = queue.Queue(maxsize=12)
    conv = O.Oracle(ofmt, args.threshold, args.fix, int(args.mode_ac), sc16q11_table_bits=args.sc16q11_table_bits)
    demo = O.Oracle(ofmt, args.threshold, args.fix, int(args.mode_ac))
    cpu_t = {}

    def pin(cpu):
        if cpu is not None:
            try:
                os.sched_setaffinity(threading.get_native_id(), {cpu})
            except OSError:
                pass

    def reader():
        pin(cpus[0] if cpus else None)
        c0 = time.thread_time()
        carry = np.zeros(pkg.capi.OVERLAP, dtype=np.uint16)
        for b in range(nb2):
            mag, lvl, pw = conv.convert(iq[b * pkg.CHUNK * bps:(b + 1) * pkg.CHUNK * bps], pkg.CHUNK)
            data = np.concatenate([carry, mag])
            carry = mag[-pkg.capi.OVERLAP:]
            q.put((b, data, lvl, pw))
        q.put(None)
        cpu_t["reader"] = time.thread_time() - c0

    pin(cpus[1] if cpus else None)
    th = threading.Thread(target=reader)
    w0, c0, nm2 = time.perf_counter(), time.thread_time(), 0
    th.start()
    while True:
        item = q.get()
        if item is None:
            break
        b, data, lvl, pw = item
        ts = b * pkg.CHUNK * 5
        nm2 += len(demo.demod_buffer(data, ts, ts // 12000, lvl, pw, cap=1 << 14))
    th.join()
    wall2, cpu_t["demod"] = time.perf_counter() - w0, time.thread_time() - c0
    two = {"value": round(nb2 * pkg.CHUNK / wall2 / 1e6, 2), "unit": "Msamples/s", "cores": 2,
           "wall_s": round(wall2, 2), "reader_thread_cpu_s": round(cpu_t["reader"], 2),
           "demod_thread_cpu_s": round(cpu_t["demod"], 2), "messages": nm2,
           "sample": "first %d buffers of the capture, one buffer per hand-over" % nb2}
    if cpus:
        two["cpus"] = list(cpus)
    if result is not None:
        result.append(two)
    return two


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the demodulator has no CPU fallback")
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    override = os.environ.get("MSD_BENCH_DEVICE_OVERRIDE")  # every rank on this device (a one-GPU box running the N>1 path)
    ngpu = torch.cuda.device_count()
    gpu_of = (lambda r: int(override)) if override not in (None, "") else (lambda r: r % max(1, ngpu))
    device_index = gpu_of(local_rank)
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    cpus_before_pinning = sorted(os.sched_getaffinity(0))  # the N-stream CPU baseline spreads over these once the ranks are done
    pinned = None if args.no_pin else pin_to_gpu_local_cpus(torch, local_rank, world, gpu_of)
    reduce_dev = dev
    # MSD_BENCH_FORCE_DIST=1: one rank takes the N > 1 branch (process group, barrier, reductions, gather), so that the RCCL
    # leg of this script has run on a one-GPU box before it runs on eight (tests/test_gpu_bench_rccl_one_rank.py)
    distributed = world > 1 or os.environ.get("MSD_BENCH_FORCE_DIST", "") not in ("", "0")
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
            reduce_dev = torch.device("cpu")

    # host threads of the buffer-parallel resolve: share the host's CPUs between the ranks of the node
    if "MSD_RESOLVE_THREADS" not in os.environ:
        os.environ["MSD_RESOLVE_THREADS"] = str(max(4, min(64, (os.cpu_count() or 8) // max(1, world))))
    pkg = graft.load_package()
    fmt = {"uc8": pkg.FMT_UC8, "sc16": pkg.FMT_SC16, "sc16q11": pkg.FMT_SC16Q11}[args.format]
    bps = 2 if fmt == pkg.FMT_UC8 else 4
    n = args.samples
    if args.batch <= 0:
        # four batches per pass keep the pipeline (four deep) full across the passes' ends; bigger launches amortise the
        # two launch gaps per batch and the scan kernel's ramp and tail (0.145 instead of 0.158 ms per 64 Mi samples)
        args.batch = max(1 << 26, min(1 << 27, n // 4))
    batch = min(args.batch, ((n + pkg.CHUNK - 1) // pkg.CHUNK) * pkg.CHUNK)
    batch = max(pkg.CHUNK, (batch // pkg.CHUNK) * pkg.CHUNK)

    # ---- synthetic capture, generated on the host from a seed, then made resident in HBM ----
    seed = pkg.sharding.capture_seed(rank)
    cfg = pkg.siggen.make_cfg(seed=seed, fmt=fmt, msgs_per_sec=args.msgs_per_sec, ac_per_sec=500 if args.mode_ac else 0,
                              noise_fs=args.noise_fs)
    t0 = time.time()
    if args.input == "random":
        iq = np.random.default_rng(seed).integers(0, 256, size=n * bps, dtype=np.uint8)
    else:
        iq = pkg.siggen.generate(cfg, n)
    gen_s = time.time() - t0
    d_iq = torch.from_numpy(iq).to(dev)
    torch.cuda.synchronize()

    stream = torch.cuda.Stream(device=dev)  # one explicit stream shared by the contexts: their kernels stay in order
    # One context per GPU.  The passes over the capture run back to back on it: msd_restart() starts the next
    # pass while the last batches of the current one are still in flight (every pass still begins with an
    # empty ICAO filter, a zero clock and zero counters, and all K passes are complete before the clock stops).
    nctx = 1
    dems = [pkg.Demodulator(fmt=fmt, preamble_threshold=args.threshold, nfix_crc=args.fix, mode_ac=int(args.mode_ac), device=device_index, dc_filter=args.dcfilter,
                            max_batch_samples=batch, stream=stream.cuda_stream, message_capacity=1 << 21,
                            decode_fields=args.fields, **({"sc16q11_table_bits": args.sc16q11_table_bits} if args.sc16q11_table_bits else {}))
            for _ in range(nctx)]
    for d in dems:
        d.set_timing_interval(args.timing_interval)
    dem = dems[0]

    DEPTH = int(os.environ.get("MSD_BENCH_DEPTH", pkg.capi.PIPELINE_DEPTH))

    def run_steps(k, collect_timing=None):
        """k passes over the capture on one context, back to back: msd_restart() lets the first batches of the
        next pass queue up behind the last ones of the current pass.  Returns the messages of the last pass."""
        d, inflight, counts = dems[0], [], {}

        def collect_one():
            cap_id, bidx = inflight.pop(0)
            got = d.collect_fields(copy=False)[0] if args.fields else d.collect(copy=False)
            counts[cap_id] = counts.get(cap_id, 0) + len(got)
            if collect_timing is not None:
                t = d.timing()
                t["batch_in_pass"] = bidx
                collect_timing.append(t)

        for s in range(k):
            d.restart()
            off = 0
            while off < n:
                if len(inflight) == DEPTH:
                    collect_one()
                m = min(batch, n - off)
                d.launch_device(d_iq.data_ptr() + off * bps, m, off + m >= n)
                inflight.append((s, off // batch))
                off += m
        while inflight:
            collect_one()
        return counts.get(k - 1, 0)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Before the W warm-up steps the contract asks for: let the machine settle.  A step is 2 ms, so W = 3..5 of them are
    # over long before a fresh box has calmed down -- the first process on one came out up to 11 % below every later
    # one (260 against 293 GS/s: same kernels, same scan time, only the host's share of the pipeline slower; a pass took
    # 2.59 ms in its first half second and 2.32 ms five seconds on, profiles/r03_settle.txt).  Single passes, untimed
    # like the warm-up itself, until three half seconds in a row are within 1 % of the best seen, ten seconds at most.
    t_settle = time.perf_counter()
    settle_log = []  # ms per pass, half a second at a time
    while time.perf_counter() - t_settle < args.settle_seconds:
        w0, passes = time.perf_counter(), 0
        while time.perf_counter() - w0 < 0.5:
            run_steps(1)
            passes += 1
        torch.cuda.synchronize()
        settle_log.append((time.perf_counter() - w0) * 1e3 / passes)
        if len(settle_log) >= 3 and max(settle_log[-3:]) <= 1.01 * min(settle_log):
            break  # three half seconds in a row within 1 % of the best so far: nothing is settling any more
    settle_log = [round(x, 3) for x in settle_log]
    run_steps(args.warmup)
    timings = []
    barrier()
    t0 = time.perf_counter()
    nmsg = run_steps(args.steps, timings)
    barrier()
    elapsed = time.perf_counter() - t0
    my_ms_per_step = elapsed * 1e3 / max(1, args.steps)
    per_rank = [{"rank": rank, "ms_per_step": my_ms_per_step, "seed": seed, "device": device_index, "messages": nmsg,
                 "cpus": (pinned or {}).get("cpu_list")}]
    if distributed:  # every rank's own time, capture and placement, beside the MAX / SUM the contract asks for
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    per_rank_ms = [float(x["ms_per_step"]) for x in per_rank]
    elapsed, nmsg_total, total_samples = pkg.sharding.reduce_job(elapsed, nmsg, n, device=reduce_dev)

    ms_per_step = elapsed * 1e3 / max(1, args.steps)
    value = total_samples / (ms_per_step * 1e-3) / 1e6  # Msamples/s, whole job

    # ---- roofline of the dominant kernel, from HIP events around one launch in --timing-interval of the timed region ----
    # (msd_timing.timed_batches counts on across msd_restart, so a launch was timed iff the count moved with its collect)
    measured, last = [], None
    for t in timings:
        if last is not None and t["timed_batches"] != last:
            measured.append(t)
        last = t["timed_batches"]
    nb = (n + batch - 1) // batch
    full = [t for t in measured if t["scan_kernel_ms"] > 0 and (t["batch_in_pass"] != nb - 1 or n % batch == 0)] or \
           [t for t in measured if t["scan_kernel_ms"] > 0]
    scan_ms = [t["scan_kernel_ms"] for t in full]
    # the first launch of a pass finds no finished predecessor whose message records it could carry:
    # it runs the instantiation without the record slice
    ms_rec = [t["scan_kernel_ms"] for t in full if t["batch_in_pass"] != 0]
    ms_alone = [t["scan_kernel_ms"] for t in full if t["batch_in_pass"] == 0]
    avg_ms = float(np.mean(scan_ms)) if scan_ms else float("nan")
    launch_samples = batch if n >= batch else n
    achieved_gbs = launch_samples * bps / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes of this same command,
    # scripts/pmc_traffic.sh; the summary is committed under profiles/).  FETCH_SIZE is doubled as
    # MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; WRITE_SIZE is taken as is.
    traffic = None
    # the latest round's measurement for this sample format; the launches of the in-order layout also carry the
    # previous batch's message records (fused), the others are the scan alone (_scan_only)
    side = (dem.flags & pkg.capi.CFG_CHAIN_SIDE_STREAMS) or (not (dem.flags & pkg.capi.CFG_CHAIN_IN_ORDER) and
                                                             (args.mode_ac or args.format != "uc8" or args.dcfilter))
    fused = not (dem.flags & pkg.capi.CFG_EMIT_KERNEL) and not args.fields and not side
    # one traffic file per workload: the Mode A/C scan also stores the magnitudes (mag_out), the 16-bit formats read 4 B per sample
    kind = ("modeac" if args.mode_ac else "") + ("" if args.format == "uc8" else "sc16" if args.format == "sc16" else "sc16q11")
    tag = "_%straffic.json" % (kind + "_" if kind else "")
    tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                    if f.endswith(tag) and (kind or not any(k in f for k in ("sc16", "modeac"))))
    tfile = os.path.join(ROOT, "profiles", tfiles[-1]) if tfiles else ""
    traffic_kernels = None
    if tfile and not args.dcfilter:
        t = json.load(open(tfile))
        sfx = "" if fused and ("FETCH_SIZE_KB_per_launch" in t) else "_scan_only"
        if t.get("samples_per_launch") == launch_samples and ("FETCH_SIZE_KB_per_launch" + sfx) in t:
            traffic = int((2 * t["FETCH_SIZE_KB_per_launch" + sfx] + t["WRITE_SIZE_KB_per_launch" + sfx]) * 1024)
            # the kernels behind the scan that pass over the batch again (Mode A/C candidates, float sums): their bytes too
            fo, wo = t.get("FETCH_SIZE_KB_per_launch_other_kernels", {}), t.get("WRITE_SIZE_KB_per_launch_other_kernels", {})
            traffic_kernels = {k: int((2 * fo.get(k, 0) + wo.get(k, 0)) * 1024) for k in sorted(set(fo) | set(wo))
                               if (2 * fo.get(k, 0) + wo.get(k, 0)) * 1024 > 0.02 * launch_samples * bps}
    roofline = {"bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_unit": "bytes per launch (PMC, %s)" % (os.path.basename(tfile) if traffic else "not measured for this workload"),
                "traffic_of_kernels_behind_the_scan": traffic_kernels,
                "algorithmic_bytes_per_launch": launch_samples * bps, "kernel": "msd_scan_kernel",
                "avg_launch_ms": round(avg_ms, 4), "samples_per_launch": launch_samples,
                "avg_launch_ms_with_records": round(float(np.mean(ms_rec)), 4) if ms_rec else None,
                "avg_launch_ms_scan_only": round(float(np.mean(ms_alone)), 4) if ms_alone else None,
                "launch_ms_min_max": [round(min(scan_ms), 4), round(max(scan_ms), 4)] if scan_ms else None,
                "algorithmic_bytes_per_sample": bps, "launches_timed": len(scan_ms), "launches": len(timings)}
    # The same fraction from the profiler's side: the newest profiles/*_scan_launches.json holds every launch
    # duration of a rocprofv3 --kernel-trace run of this workload (scripts/r6_profiles.sh; >= 30 launches, the first launch of
    # each instantiation excluded by count).  The HIP-event figure above is un-profiled; the profiler's launches are 5-7 %
    # longer (its own per-dispatch work); the line carries both.
    ltag = "_%sscan_launches.json" % (kind + "_" if kind else "")
    lfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                    if f.endswith(ltag) and (kind or not any(k in f for k in ("sc16", "modeac"))))
    roofline["frac_rocprof"] = None
    if lfiles and not args.dcfilter:
        L = json.load(open(os.path.join(ROOT, "profiles", lfiles[-1])))
        if L.get("samples_per_launch") == launch_samples and L.get("launches_kept"):
            gbs = lambda us: launch_samples * bps / (us * 1e-6) / 1e9  # noqa: E731
            roofline["frac_rocprof"] = {"mean": round(gbs(L["mean_us"]) / HBM_PEAK_GBS, 4), "median": round(gbs(L["median_us"]) / HBM_PEAK_GBS, 4),
                                        "mean_us": L["mean_us"], "median_us": L["median_us"], "launches": L["launches_kept"],
                                        "excluded": L.get("excluded"), "source": "profiles/" + lfiles[-1]}
    # what the kernel is really bound by (it is not HBM): from the latest profiles/*_binding.json, a summary of SQ counter
    # passes of this same command (scripts/r3_profiles.sh) and of the issue-rate microbenchmark
    bfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_binding.json"))
    if bfiles and args.format == "uc8" and not args.mode_ac:
        binding = json.load(open(os.path.join(ROOT, "profiles", bfiles[-1])))
        roofline["binding"] = binding
        # the three ceilings side by side: HBM (frac above), vector-ALU issue, the table gathers of the conversion alone
        roofline["valu_lane_util"] = binding.get("valu_lane_util")
        roofline["conversion_floor_ms"] = binding.get("conversion_floor_ms")
        floor = (binding.get("conversion_floor_ms") or {}).get("value")
        roofline["ceilings"] = {"hbm": roofline["frac"], "valu_issue": binding.get("frac"),
                                "conversion_only_over_full_kernel": round(floor / avg_ms, 3) if floor and avg_ms == avg_ms else None,
                                "what": "share of the launch explained by each limit: algorithmic bytes / HBM peak; vector-ALU issue cycles of the "
                                        "measured instruction mix / kernel cycles; the kernel stopped after the conversion (one table gather per "
                                        "sample, nothing else) / the full kernel"}
    # in the in-order layout without field decoding the scan's wavefronts also write the previous batch's message
    # records; MSD_EMIT_FUSED=0 gives them a kernel of their own and times the scan alone
    roofline["launch_includes"] = ("the previous batch's message records (35 000 x 56 B to host memory); "
                                   "MSD_CFG_EMIT_KERNEL times the scan alone") if fused else "the scan only"

    out = {
        "metric": "IQ Msamples/s, 2.4 MSPS %s, Mode S demodulation (CRC-valid msgs/s alongside)" % args.format.upper(),
        "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32 (u8 IQ -> u16 magnitude -> int32 correlators, 24-bit CRC)", "data": "synthetic",
        "config": {"workload": "%.3f GiB synthetic 2.4 MSPS %s capture per GPU, %s, %s, preamble threshold %d, "
                               "%s, seeds 10901+rank" % (n * bps / 2**30, args.format.upper(),
                                                         "Mode S + Mode A/C" if args.mode_ac else "Mode S only",
                                                         "--no-fix" if args.fix == 0 else "--fix", args.threshold,
                                                         "uniformly random bytes" if args.input == "random" else
                                                         "%d frames/s%s" % (args.msgs_per_sec, "" if args.noise_fs == 0.02
                                                                            else ", noise sigma %.3f FS" % args.noise_fs)),
                   "samples_per_gpu": n, "batch_samples": batch, "parallelism": "independent capture per GPU, no collective",
                   "captures": "one context per GPU; passes over the capture back to back (msd_restart), each starting "
                               "from an empty ICAO filter, all complete inside the timed region"},
        "msgs_per_s": round(nmsg_total / (ms_per_step * 1e-3), 1), "messages_per_step": nmsg_total,
        "signal_seconds_per_wall_second": round(value * 1e6 / 2.4e6, 1),
        "roofline": roofline,
        "pipeline_ms": ({**{k: round(float(np.mean([t[k] for t in measured])), 4) for k in
                            ("scan_kernel_ms", "other_kernels_ms")},
                         **{k: round(float(np.mean([t[k] for t in timings])), 4) for k in
                            ("d2h_ms", "resolve_ms", "hits", "tries")},
                         "reruns": int(max(t["reruns"] for t in timings)),
                         "resolve_fallback": int(max(t["resolve_fallback"] for t in timings))} if timings and measured else None),
        "capture_generation_s": round(gen_s, 2),
        "per_rank_ms_per_step": {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3)},
        "settle_ms_per_pass": settle_log,  # untimed single passes before the warm-up, by half second (rank 0)
        "host_placement": ({k: v for k, v in pinned.items() if k != "cpu_list"} if pinned else "process affinity left as found"),
        "ranks": [{k: (("%d-%d" % (v[0], v[-1])) if k == "cpus" and v else v) for k, v in r.items()} for r in per_rank],
        "dist_backend": args.dist_backend if distributed else None,
        "host_threads_per_rank": "2 busy (caller: polls events, replays filter changes, queues kernels; helper: copies the "
                                 "records, power statistics) + an idle pool for the host resolver",
        "resolve_stage": ("gpu, %.2f passes per batch, %d batches handed to the host resolver"
                          % (float(np.mean([t["resolve_passes"] for t in timings])), int(timings[-1]["resolve_fallback"]))
                          if timings and any(t["resolve_passes"] for t in timings)
                          else "host threads (%s)" % os.environ["MSD_RESOLVE_THREADS"]),
    }

    if args.dcfilter:
        # how the DC block of the last batch was computed (msd_dc_filter_status): by the exact parallel-in-time kernels, in how many passes
        ex, passes, guessed, blocks = dems[0].dc_filter_status()
        out["dc_block"] = {"exact_parallel": bool(ex), "passes": passes, "blocks": blocks, "blocks_guessed_all_passes": guessed,
                           "what": "the last batch's DC block (convert.c:137-138): blocks evaluated from 64 candidate start states per pass, "
                                   "an in-order walk that is exact by table hit or by monotonicity; false = the in-order kernel behind the passes took the batch"}
    # ---- N > 1: the evidence an N = 1 line carries, for every rank, after the clock stopped ----
    if world > 1 and not args.no_check:
        # every rank: its GPU's ordered message list and counters over the first 256 buffers of ITS capture (seed 10901 + rank)
        # against the oracle on the same samples
        O = graft.load_oracle()
        ofmt = {"uc8": O.FMT_UC8, "sc16": O.FMT_SC16, "sc16q11": O.FMT_SC16Q11}[args.format]
        head = min(n, 256 * pkg.CHUNK)
        t0 = time.perf_counter()
        hwant, hwstats = O.Oracle(ofmt, args.threshold, args.fix, int(args.mode_ac), dc_filter=args.dcfilter,
                                  sc16q11_table_bits=args.sc16q11_table_bits).replay(iq[: head * bps], cap=1 << 21)
        oracle_s = time.perf_counter() - t0
        dem.reset()
        hgot = pkg.replay_device(dem, d_iq.data_ptr(), head, batch)
        mine = {"rank": rank, "seed": seed, "buffers": (head + pkg.CHUNK - 1) // pkg.CHUNK, "messages": int(len(hwant)),
                "diff": diff_against_oracle(hgot, dem.stats(), hwant, hwstats),
                "oracle_msamples_per_s_one_core": round(head / oracle_s / 1e6, 1)}
        every = [None] * world
        dist.all_gather_object(every, mine)
        every = sorted(every, key=lambda x: x["rank"])
        out["message_set_diff_vs_oracle_per_rank"] = every
        out["message_set_diff_vs_oracle"] = int(sum(x["diff"] for x in every))
        out["messages_checked"] = int(sum(x["messages"] for x in every))
        if out["message_set_diff_vs_oracle"]:
            raise SystemExit("bench: GPU messages differ from the oracle on some rank: " + json.dumps(every))
    if world > 1 and rank == 0 and not args.no_cpu_baseline and not args.dcfilter:
        # SURVEY.md 8(d) CPU baseline, form (b): N independent streams (captures 10901 .. 10901 + N - 1) on 2 N pinned cores,
        # each stream the reference's reader thread + demodulator thread (readsb.c:271-285, 820-855), thread CPU time
        # (CLOCK_THREAD_CPUTIME_ID, util.c:102-115) and wall time.  The other ranks' GPU work is over: the host's cores are free.
        import threading
        O = graft.load_oracle()
        ofmt = {"uc8": O.FMT_UC8, "sc16": O.FMT_SC16, "sc16q11": O.FMT_SC16Q11}[args.format]
        nb2 = min(n, 1 << 27) // pkg.CHUNK
        streams_iq = [iq] + [pkg.siggen.generate(pkg.siggen.make_cfg(seed=pkg.sharding.capture_seed(r), fmt=fmt, msgs_per_sec=args.msgs_per_sec,
                                                                     ac_per_sec=500 if args.mode_ac else 0, noise_fs=args.noise_fs),
                                                 nb2 * pkg.CHUNK) for r in range(1, world)]
        avail = cpus_before_pinning if len(cpus_before_pinning) >= 2 * world else sorted(os.sched_getaffinity(0))
        pairs = [(avail[2 * r], avail[2 * r + 1]) if len(avail) >= 2 * world else None for r in range(world)]
        results, threads = [[] for _ in range(world)], []
        w0 = time.perf_counter()
        for r in range(world):
            th = threading.Thread(target=two_thread_stream, args=(O, pkg, streams_iq[r], bps, ofmt, args, nb2, pairs[r], results[r]))
            th.start()
            threads.append(th)
        for th in threads:
            th.join()
        wall = time.perf_counter() - w0
        per_stream = [x[0] for x in results]
        out["cpu_baseline"] = {"value": round(world * nb2 * pkg.CHUNK / wall / 1e6, 2), "unit": "Msamples/s", "cores": 2 * world, "kind": "port",
                               "sample": "%d independent streams (captures of seeds 10901..%d), the first %d buffers (%.2f GiB) of each, "
                                         "each stream a reader thread (IQ -> magnitude) and a demodulator thread as in readsb.c:271-285,"
                                         "820-855, every thread pinned to a CPU of its own" % (world, 10900 + world, nb2, nb2 * pkg.CHUNK * bps / 2**30),
                               "wall_s": round(wall, 2),
                               "thread_cpu_s": {"readers": round(sum(x["reader_thread_cpu_s"] for x in per_stream), 2),
                                                "demodulators": round(sum(x["demod_thread_cpu_s"] for x in per_stream), 2)},
                               "host": "%d logical CPUs, %s" % (os.cpu_count() or 0, cpu_model()),
                               "streams": per_stream}
        os.sched_setaffinity(0, set(avail))

    # ---- CPU baseline: the oracle on this host, bounded sample (rank 0, N=1 only), after the clock stopped ----
    want = wstats = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        O = graft.load_oracle()
        ns = min(n, args.cpu_sample)
        ofmt = {"uc8": O.FMT_UC8, "sc16": O.FMT_SC16, "sc16q11": O.FMT_SC16Q11}[args.format]
        # (a) one thread doing everything (orc_replay: IQ -> magnitude -> demodulator), whole passes for ~10 s
        passes, cpu_s = 0, 0.0
        while passes < 1 or (cpu_s < 10.0 and passes < 8):
            orc = O.Oracle(ofmt, args.threshold, args.fix, int(args.mode_ac), dc_filter=args.dcfilter, sc16q11_table_bits=args.sc16q11_table_bits)
            t0 = time.perf_counter()
            w, ws = orc.replay(iq[: ns * bps], cap=1 << 21)
            cpu_s += time.perf_counter() - t0
            passes += 1
        cpu_s /= passes
        if ns == n:
            want, wstats = w, ws
        # (b) the reference's own two-thread structure, one stream (two_thread_stream above)
        two = None
        if not args.dcfilter:
            two = two_thread_stream(O, pkg, iq, bps, ofmt, args, min(ns, 1 << 28) // pkg.CHUNK)
        out["cpu_baseline"] = {"value": round(ns / cpu_s / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
                               "sample": "first %d samples (%.2f GiB) of the same capture, oracle replay incl. IQ->magnitude, "
                                         "%d passes of %.1f s wall" % (ns, ns * bps / 2**30, passes, cpu_s),
                               "msgs_per_s": round(len(w) / cpu_s, 1),
                               "host": "%d logical CPUs, %s" % (os.cpu_count() or 0, cpu_model()),
                               "two_threads_like_the_reference": two}
    if rank == 0 and world == 1 and not args.no_check and (args.check or want is not None):
        # the whole capture, message for message and counter for counter, against the oracle
        O = graft.load_oracle()
        if want is None:
            ofmt = {"uc8": O.FMT_UC8, "sc16": O.FMT_SC16, "sc16q11": O.FMT_SC16Q11}[args.format]
            want, wstats = O.Oracle(ofmt, args.threshold, args.fix, int(args.mode_ac), dc_filter=args.dcfilter,
                                    sc16q11_table_bits=args.sc16q11_table_bits).replay(iq, cap=1 << 21)
        dem.reset()
        got = pkg.replay_device(dem, d_iq.data_ptr(), n, batch)
        gstats = dem.stats()
        ndiff = diff_against_oracle(got, gstats, want, wstats)
        out["message_set_diff_vs_oracle"] = ndiff
        out["messages_checked"] = int(len(want))
        if ndiff:
            raise SystemExit("bench: GPU messages differ from the oracle: " + json.dumps(out))
        # ---- a third witness on the head of the capture: the second reading of the reference (tests/indep_demod.py, numpy /
        # plain Python, written from the reference's sources without the oracle), against the GPU's list directly.  Reported,
        # never fatal: the oracle above is the gate. ----
        if not args.sc16q11_table_bits:
            try:
                tdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests")
                if tdir not in sys.path:
                    sys.path.insert(0, tdir)
                import indep_demod
                nb = 128
                head = min(n, nb * pkg.CHUNK + 4096)
                w0 = time.perf_counter()
                smsgs, _ = indep_demod.Receiver(args.format, args.threshold, args.fix, bool(args.mode_ac),
                                                dc_filter=args.dcfilter).replay(iq[: head * bps].tobytes())
                # whole buffers only: the short last buffer of the head sees another end of the stream than the capture's
                limit = (head // pkg.CHUNK) * pkg.CHUNK * 5
                smsgs = [x for x in smsgs if x["timestampMsg"] < limit]
                ghead = got[got["timestampMsg"] < limit]
                sdiff = abs(len(smsgs) - len(ghead))
                for x, y in zip(smsgs, ghead):
                    nbytes = int(y["msgbits"]) // 8
                    same = (x["timestampMsg"] == int(y["timestampMsg"]) and x["msgtype"] == int(y["msgtype"]) and
                            x["addr"] == int(y["addr"]) and x["correctedbits"] == int(y["correctedbits"]) and
                            x["msg"][:nbytes] == bytes(y["msg"][:nbytes]))
                    if same and x["msgtype"] != 32:
                        same = (x["score"] == int(y["score"]) and x["bestphase"] == int(y["bestphase"]) and
                                x["crc"] == int(y["crc"]) and x["signalLevel"] == float(y["signalLevel"]))
                    sdiff += 0 if same else 1
                out["message_set_diff_vs_second_reading"] = {
                    "diff": sdiff, "messages": len(smsgs), "buffers": head // pkg.CHUNK, "seconds": round(time.perf_counter() - w0, 1),
                    "what": "tests/indep_demod.py on the first buffers of the same capture, compared with the GPU's list (not the "
                            "oracle's): timestamps, bytes, address, repaired bits, score, phase, CRC, signal level as the same double"}
            except Exception as e:  # noqa: BLE001
                out["message_set_diff_vs_second_reading"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.dcfilter:
        # ---- the drop-in figure: the same capture handed over in host memory (what --ifile sees, sdr_ifile.c:192-216) -- upload
        # over PCIe + kernels + records home, batches in flight, page-locked buffers.  Reported beside `value`, never as it. ----
        try:
            pb = min(batch, 1 << 26)
            nbuf = (n + pb - 1) // pb
            bufs = [dem.host_buffer(pb * bps) for _ in range(nbuf)]
            for i, b in enumerate(bufs):
                m = min(pb, n - i * pb)
                b[: m * bps] = iq[i * pb * bps:(i * pb + m) * bps]
            best = None
            for rep in range(3):
                dem.reset()
                torch.cuda.synchronize()
                t0, inflight, nm = time.perf_counter(), 0, 0
                for i, b in enumerate(bufs):
                    m = min(pb, n - i * pb)
                    if inflight == DEPTH:
                        nm += len(dem.collect(copy=False))
                        inflight -= 1
                    dem.launch_host(b, m, last=i == nbuf - 1)
                    inflight += 1
                while inflight:
                    nm += len(dem.collect(copy=False))
                    inflight -= 1
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out["pcie_inclusive"] = {"value": round(n / best / 1e6, 1), "unit": "Msamples/s", "link_GBps": round(n * bps / best / 1e9, 1),
                                     "messages": nm, "batch_samples": pb,
                                     "what": "msd_launch_host + msd_collect from page-locked host buffers, %d batches in flight, "
                                             "best of 3 passes over the same capture; PCIe Gen5 x16 = 63 GB/s" % DEPTH}
            del bufs
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline line
            out["pcie_inclusive"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.dcfilter:
        # ---- one lone capture, start to finish: the steady-state figure above is K passes chained by msd_restart(); a single
        # file costs the pipeline's fill and drain as well ----
        try:
            lone = []
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_steps(1)
                torch.cuda.synchronize()
                lone.append((time.perf_counter() - t0) * 1e3)
            out["single_capture_ms"] = {"value": round(min(lone), 3), "all": [round(x, 3) for x in lone], "samples": n,
                                        "msamples_per_s": round(n / min(lone) / 1e3, 1),
                                        "what": "msd_restart, every batch of the capture launched and collected, device idle before and after"}
        except Exception as e:  # noqa: BLE001
            out["single_capture_ms"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_dropin and args.format == "uc8" and not args.dcfilter:
        # ---- the path north_star names: "keeping the sdr.h / fifo.h mag_buf interface so it drops in behind --ifile" -- the
        # replay tool (readsb's part: option keys, hooks, the ifile handler) on a 10 s capture file (BASELINE configs[0]: 24 M
        # samples), once through the literal mag_buf path (iq_convert_fn per 131072-sample block, FIFO, demodulate2400(struct
        # mag_buf *) on the consumer thread; readsb.c:820-855, sdr_ifile.c:164-237), once through the fused path, and the
        # mag_buf path again in real time (--throttle, sdr_ifile.c:218-226): per-buffer latency and missed deadlines ----
        import subprocess
        import tempfile
        exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "msd_replay")
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        cap = os.path.join(tmpdir, "msd_bench_capture_%d.uc8" % os.getpid())
        dropin = {"capture": "first %d samples (10 s at 2.4 MSPS) of the benchmark capture, as a file" % min(n, 24_000_000)}
        try:
            iq[: 2 * min(n, 24_000_000)].tofile(cap)
            base_args = [exe, "--ifile", cap, "--iformat", "uc8", "--no-output", "--timing", "--device", str(device_index),
                         "--preamble-threshold", str(args.threshold)] + (["--fix"] if args.fix == 1 else ["--aggressive"] if args.fix == 2 else ["--no-fix"]) + \
                        (["--modeac"] if args.mode_ac else [])

            def replay(extra, timeout):
                best = None
                for rep in range(1 if "--throttle" in extra else 3):
                    res = subprocess.run(base_args + extra, capture_output=True, text=True, timeout=timeout)
                    line = [l for l in res.stderr.splitlines() if l.startswith("{")]
                    if res.returncode != 0 or not line:
                        return {"error": (res.stderr or "no timing line")[-300:]}
                    d = json.loads(line[-1])
                    if best is None or d["wall_s"] < best["wall_s"]:
                        best = d
                return best
            m = replay(["--path", "magbuf"], 600)
            dropin["magbuf"] = m if "error" in m else {
                "value": m["msamples_per_s"], "unit": "Msamples/s", "wall_s": m["wall_s"], "buffers": m["buffers"], "messages": m["messages"],
                "demodulate2400_us_per_buffer": {"p50": m["demod_us_p50"], "p99": m["demod_us_p99"], "max": m["demod_us_max"]},
                "iq_convert_fn_us_per_buffer": {"p50": m["convert_us_p50"], "p99": m["convert_us_p99"]},
                "what": "msd_replay --path magbuf: reader thread (read, iq_convert_fn on the GPU, fifo_enqueue) + consumer thread "
                        "(fifo_dequeue, demodulate2400(struct mag_buf *) on the GPU, fifo_release), best of 3"}
            f = replay(["--path", "fused"], 600)
            dropin["fused"] = f if "error" in f else {"value": f["msamples_per_s"], "unit": "Msamples/s", "wall_s": f["wall_s"], "messages": f["messages"],
                                                      "what": "msd_replay --path fused: the same handler, 64 buffers per msd_launch_host, best of 3"}
            # the same two paths over 1024 buffers: the per-buffer cost without the run's first calls
            if n >= 1 << 27:
                iq[: 2 << 27].tofile(cap)
                ms_, fs_ = replay(["--path", "magbuf"], 600), replay(["--path", "fused"], 600)
                dropin["steady_state_1024_buffers"] = {
                    "magbuf": ms_ if "error" in ms_ else {"value": ms_["msamples_per_s"], "unit": "Msamples/s", "wall_s": ms_["wall_s"],
                                                          "demodulate2400_us_per_buffer_p50": ms_["demod_us_p50"],
                                                          "iq_convert_fn_us_per_buffer_p50": ms_["convert_us_p50"]},
                    "fused": fs_ if "error" in fs_ else {"value": fs_["msamples_per_s"], "unit": "Msamples/s", "wall_s": fs_["wall_s"]}}
                iq[: 2 * min(n, 24_000_000)].tofile(cap)
            t = replay(["--path", "magbuf", "--throttle"], 120)
            dropin["magbuf_throttle"] = t if "error" in t else {
                "wall_s": t["wall_s"], "buffers": t["buffers"], "deadline_misses": t["deadline_misses"], "buffer_period_ms": round(131072 / 2400.0, 2),
                "release_to_messages_us": {"p50": t["latency_us_p50"], "p99": t["latency_us_p99"], "max": t["latency_us_max"]},
                "what": "the same path paced like a live receiver (--throttle): from fifo_enqueue of a buffer to its last message at the sink"}
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline line
            dropin["error"] = repr(e)[:200]
        finally:
            try:
                os.unlink(cap)
            except OSError:
                pass
        out["dropin_magbuf"] = dropin
    if rank == 0 and world == 1 and not args.no_also and not args.no_cpu_baseline and args.format == "uc8" and \
            not (args.mode_ac or args.fields or args.dcfilter or args.fix) and n == 1 << 29:
        # BASELINE configs[2] and configs[4] at full size, after the clock stopped, each against the oracle
        for d in dems:
            d.close() if hasattr(d, "close") else None
        out["also"] = [run_also(["--format", "sc16", "--samples", str(1 << 28)]), run_also(["--mode-ac", "--fix", "1"]),
                       # round 6: --dcfilter (convert.c:113-163), whose DC block is exact and parallel in time now (0.13 GS/s in order)
                       run_also(["--dcfilter", "--samples", str(1 << 27)])]
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
