#!/usr/bin/env python3
"""bench.py -- stereo frame-pairs/sec (detect + quad match), KITTI 1241x376.

One "step" = one pass of the hot path over one batch of synthetic input: every
camera stream of this rank's stream group receives one new stereo pair
(pushBack: detect on 2 images) and is quad-matched against its previous pair
(matchFeatures(2)).  Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S]

N>1 is launched by the driver through torch.distributed.run, one rank per GPU;
ranks run independent streams (no data-path collective: the path partitions by
camera sequence) and only the timing is reduced (MAX over ranks).

Rank 0 prints ONE JSON line; see README/DESIGN.md for the `roofline` and
`cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

METRIC = "stereo frame-pairs/sec (detect+match), KITTI 1241x376; matches bit-exact"
W, H = 1241, 376
# BASELINE.json configs: the metric is quoted on "kitti" (cfg-2, the default and the only
# driver-run line); "1080p" (cfg-3) and "4k" (cfg-5) are context runs for DESIGN.md
WORKLOADS = {
    "kitti": dict(W=1241, H=376, params={}, streams=256, cap=32768,
                  label="KITTI 1241x376 stereo quad-match (prev/curr x L/R), default 50x50 bins"),
    "1080p": dict(W=1920, H=1080, params={}, streams=48, cap=131072,
                  label="1920x1080 stereo quad-match, default parameters"),
    "4k": dict(W=3840, H=2160, params={"nms_n": 3, "match_binsize": 25}, streams=12, cap=524287,
               label="3840x2160 stereo quad-match, nms_n=3, 25x25 bins"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
KERNELS = ("detect_nms", "emit_features", "bin_hist", "bin_scan", "bin_fill", "bin_sort",
           "match", "chain", "emit_matches")


NOISE = 0  # --noise: grey levels of uniform sensor noise added to every frame (context runs only)
NOISE_FRAC = 1.0  # --noise-frac: share of the pixels that receive it


def make_frames(pkg, n_streams: int, n_frames: int, rank: int, size=None, noise=None, noise_frac=None, n_seeds: int = 8):
    """[T][2][S][H][bpl] uint8: stream s follows sequence (seed 1 + (rank*S+s) % n_seeds),
    phase-shifted so that no two streams of a rank see identical frames."""
    w, h = size if size else (W, H)
    noise = NOISE if noise is None else noise
    noise_frac = NOISE_FRAC if noise_frac is None else noise_frac
    bpl = pkg.synth.bytes_per_line(w)
    out = np.zeros((n_frames, 2, n_streams, h, bpl), np.uint8)
    cache = {}  # (seed, k) -> (left, right): streams n_seeds apart share the sequence, shifted in time

    def pair(seed, k):
        if (seed, k) not in cache:
            dx, dy = (5 * k) % 20, k % 20
            pr = [pkg.synth.frame(w, h, dx, dy, 8, 1, seed), pkg.synth.frame(w, h, dx + 12, dy, 8, 1, seed)]
            if noise:
                rng = np.random.default_rng(seed * 1000 + k)
                for i_ in range(2):
                    nz = rng.integers(-noise, noise + 1, pr[i_].shape)
                    if noise_frac < 1.0:
                        nz = nz * (rng.random(pr[i_].shape) < noise_frac)
                    im = np.clip(pr[i_].astype(np.int32) + nz, 0, 255).astype(np.uint8)
                    im[:, w:] = 0
                    pr[i_] = im
            cache[(seed, k)] = tuple(pr)
        return cache[(seed, k)]

    for s, (gs, seed, phase) in enumerate(stream_assignment(rank, n_streams, n_seeds)):
        for t in range(n_frames):
            out[t, 0, s], out[t, 1, s] = pair(seed, t + phase)
    return out, bpl


def cpu_baseline(ob, frames, dims, budget_s: float, over=None):
    """The oracle (CPU port of the reference SSE path; flow bit-identical to the
    reference, quad per SURVEY App. A.7) on a bounded sample of the SAME
    workload: stream 0's consecutive stereo pairs, one thread."""
    o = ob.Oracle()
    p = ob.Params.default(**(over or {}))
    T = frames.shape[0]
    prev = None
    pairs = 0
    results = []
    t0 = time.perf_counter()
    k = 0
    while True:
        t = k % T
        cur = [o.compute_features(p, frames[t, c, 0], dims)[1] for c in (0, 1)]
        if prev is not None:
            results.append(o.matching(p, dims, 2, prev[0], prev[1], cur[0], cur[1]))
            pairs += 1
        prev = cur
        k += 1
        if pairs >= 3 and time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    # the first detect has no match: charge detect+match per pair
    return pairs / dt, pairs, dt, results


def cpu_baseline_threads(ob, frames, dims, budget_s: float, over, n_threads: int):
    """The same CPU path on n_threads host threads, one camera stream each (the
    C calls release the GIL): what the host of this GPU does on the workload when
    every core of its share is put on it."""
    import threading
    T = frames.shape[0]
    done = [0] * n_threads
    t_end = time.perf_counter() + budget_s

    def work(i):
        o = ob.Oracle()
        p = ob.Params.default(**(over or {}))
        s = i % frames.shape[2]
        prev, k = None, 0
        while time.perf_counter() < t_end or done[i] < 1:
            t = k % T
            cur = [o.compute_features(p, frames[t, c, s], dims)[1] for c in (0, 1)]
            if prev is not None:
                o.matching(p, dims, 2, prev[0], prev[1], cur[0], cur[1])
                done[i] += 1
            prev = cur
            k += 1

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    return sum(done) / dt, sum(done), dt


def in_window_pairs(q, c, radius: int) -> int:
    """(query, candidate) pairs findMatch must evaluate for a flow pass: same
    class, |du| <= radius, |dv| <= radius (src/matcher.cpp:237-249)."""
    total = 0
    for cls in range(4):
        a, b = q[q[:, 3] == cls], c[c[:, 3] == cls]
        if len(a) and len(b):
            du = np.abs(a[:, None, 0] - b[None, :, 0]) <= radius
            dv = np.abs(a[:, None, 1] - b[None, :, 1]) <= radius
            total += int(np.count_nonzero(du & dv))
    return total


def nm_max_hint(grp) -> int:
    """Longest match list of the last step (sizes the download slot of the post stage)."""
    _, nm = grp.getCounts()
    return int(nm.max()) if len(nm) else 0


def stream_assignment(rank: int, n_streams: int, n_seeds: int = 8):
    """Global stream ids owned by `rank` and their (seed, phase): streams are
    independent camera sequences, sharded rank-major with no overlap."""
    out = []
    for s in range(n_streams):
        gs = rank * n_streams + s
        out.append((gs, 1 + gs % n_seeds, gs // n_seeds))
    return out


class RankProtocol:
    """The multi-rank control flow of the bench, backend-agnostic: process group over RCCL ("nccl", one rank per GPU) or,
    for the CPU rehearsal, gloo.  Ranks never exchange data -- the path partitions by camera sequence -- only the
    barrier, the gathered rank descriptions, the number of blocks and the MAX-reduced block time cross ranks."""

    def __init__(self, backend, rank, local_rank, world, force=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.local_rank, self.world, self.backend = rank, local_rank, world, backend
        self.active = world > 1 or force
        self.dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
        if self.active:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:  # (only a single-process rehearsal lacks it; torch.distributed.run sets it)
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=self.dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier(self):
        if self.active:
            if self.backend == "nccl":
                self.dist.barrier(device_ids=[self.local_rank])
            else:
                self.dist.barrier()

    def gather(self, me):
        """Every rank's description, on every rank (rank 0 checks them)."""
        if not self.active:
            return [me]
        out = [None] * self.world
        self.dist.all_gather_object(out, me)
        return out

    def max_over_ranks(self, x, dtype=None):
        t = self.torch.tensor([x], dtype=dtype or self.torch.float64, device=self.dev)
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    def timed_blocks(self, step, fence, steps, blocks):
        """Blocks of EXACTLY `steps` steps, each between two fences, each block's time the MAX over ranks; as many
        blocks as make the timed region >= 0.5 s (at least 3, the same number on every rank) unless `blocks` says so."""
        def block():
            t0 = time.perf_counter()
            for _ in range(steps):  # the timed region: no per-kernel events in it
                step()
            fence()
            return float(self.max_over_ranks(time.perf_counter() - t0))
        block_s = [block()]
        n_blocks = blocks if blocks > 0 else max(3, min(50, int(np.ceil(0.5 / max(block_s[0], 1e-6)))))
        n_blocks = int(self.max_over_ranks(n_blocks, self.torch.int64))  # every rank must run the same number of blocks
        while len(block_s) < n_blocks:
            block_s.append(block())
        return block_s

    def close(self):
        if self.active:
            self.barrier()
            self.dist.destroy_process_group()


def check_ranks(ranks_seen, world):
    devs = {(r["pci_bus_id"], r["pci_device_id"], r["uuid"]) for r in ranks_seen}
    assert len(ranks_seen) == world and sorted(r["rank"] for r in ranks_seen) == list(range(world)), f"ranks missing: {ranks_seen}"
    assert len(devs) == world or (len(devs) == 1 and world == 1), f"ranks share a GPU: {ranks_seen}"


def dist_selftest(args):
    """world_size>1 on CPU (gloo): main()'s own control flow -- RankProtocol: process group, gathered rank
    descriptions, fence, timed blocks with the MAX-reduced clock and the agreed block count, final barrier -- around a
    stubbed step (rank r sleeps r+1 ms per step, so the slowest rank must set the clock) and the real sharding."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    proto = RankProtocol("gloo", rank, local_rank, world)
    pkg = entry.load_package()
    S = args.streams
    mine = stream_assignment(rank, S)
    # tiny frames of this rank's streams: the sharding must give different data per rank
    digest = 0
    for gs, seed, phase in mine[:2]:
        digest ^= pkg.synth.fnv1a64(pkg.synth.frame(64, 32, phase % 20, 0, 2, 1, seed)[:4])
    me = {"rank": rank, "local_rank": local_rank, "device": local_rank, "name": "cpu-rehearsal", "pci_bus_id": local_rank, "pci_device_id": 0,
          "uuid": f"fake-{local_rank}", "stream_ids": [m[0] for m in mine], "digest": digest & 0x7FFFFFFFFFFFFFFF}
    ranks_seen = proto.gather(me)
    if rank == 0:
        check_ranks(ranks_seen, world)

    def step():
        time.sleep(0.001 * (rank + 1))

    def fence():
        proto.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    block_s = proto.timed_blocks(step, fence, args.steps, args.blocks)
    dt = float(np.median(block_s))
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "streams_per_gpu": S, "steps": args.steps,
                          "global_stream_ids": [i for r in sorted(ranks_seen, key=lambda r_: r_["rank"]) for i in r["stream_ids"]],
                          "distinct_data": len({r["digest"] for r in ranks_seen}) == world,
                          "ranks_seen": ranks_seen, "blocks": {"count": len(block_s), "seconds_each": block_s},
                          "ms_per_step": 1e3 * dt / args.steps, "value": world * S * args.steps / dt}), flush=True)
    proto.close()


def context_workload(pkg, torch, dev, local_rank, name, noise, ob=None, steps=40, blocks=1, warmup=8):
    """A short run of another BASELINE.json config (or of the KITTI config with sensor noise) for the default line's
    `other_workloads` key: one block of 40 steps (three blocks of 12 paid the pipeline's fill three times: 15.3 k instead of
    18.5 k pairs/s at 1080p), the search kernel's share of the HBM yardstick from HIP events, and stream 0's last step
    checked against the oracle.  Context, never `value`; the full-length runs of the same workloads are `--workload` / `--noise`."""
    wl = WORKLOADS[name]
    w, h, S, T = wl["W"], wl["H"], wl["streams"], 3
    t_all = time.perf_counter()
    frames_np, bpl = make_frames(pkg, S, T, 0, size=(w, h), noise=noise, noise_frac=1.0, n_seeds=8 if name == "kitti" else 2)
    dims, stride = [w, h, bpl], h * bpl
    frames = torch.from_numpy(frames_np).to(dev)
    params = pkg.Params.default(**wl["params"])
    grp = pkg.StreamGroup(S, params, device=local_rank, max_features=wl["cap"], max_matches=wl["cap"])
    grp.setStream(torch.cuda.current_stream().cuda_stream)
    k = 0

    def step():
        nonlocal k
        grp.pushBackDevice(frames[k % T, 0].data_ptr(), frames[k % T, 1].data_ptr(), stride, dims, False)
        grp.matchFeatures(pkg.METHOD_QUAD)
        k += 1

    def sync():
        grp.synchronize(); torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    sync()
    bs = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        bs.append(time.perf_counter() - t0)
    dt = float(np.median(bs))
    grp.profileReset(); grp.profileEnable(True)
    for _ in range(steps):
        step()
    sync()
    grp.profileEnable(False)
    ms, n_ = grp.profileRead("match")
    nf, nm = grp.getCounts()
    nfm = nf.astype(np.float64).mean(axis=0)
    B_pair = 2 * bpl * h + 48 * (nfm[2] + nfm[3]) + 48 * nfm.sum() + 48 * float(nm.mean())
    spec, redo = grp.searchStats()
    got0 = grp.getMatches(0)
    last = (k - 1) % T
    grp.close()
    del frames
    out = {"workload": wl["label"] + (f" + noise +-{noise}" if noise else ""), "noise": noise, "value": S * steps / dt, "unit": "pairs/s",
           "ms_per_step": 1e3 * dt / steps, "streams": S, "steps": steps, "blocks_s": [round(b, 5) for b in bs],
           "features_per_image": float(nfm.mean()), "matches_per_pair": float(nm.mean()),
           "search_loop": "speculative" if spec else "tested", "queries_searched_again": redo}
    if n_:
        sec = 1e-3 * ms / n_
        out["roofline"] = {"bound": "hbm", "kernel": "match", "achieved": S * B_pair / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": S * B_pair / sec / 1e9 / HBM_PEAK_GBS, "us_per_launch": 1e6 * sec}
    if ob is not None:  # the oracle as checker: stream 0's last step
        o = ob.Oracle(); p = ob.Params.default(**wl["params"])
        f = [o.compute_features(p, frames_np[t_, c, 0], dims)[1] for t_ in ((last - 1) % T, last) for c in (0, 1)]
        out["parity_checked"] = bool(got0.tobytes() == o.matching(p, dims, 2, *f).tobytes())
    out["wall_s"] = round(time.perf_counter() - t_all, 1)
    return out


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` outside a launcher: this process -- which has imported neither torch nor the HIP
    library and never will -- starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as
    a CHILD (no exec: a process that initialised the GPU must not be replaced, and this way the rule cannot be broken by
    a later edit either), relays rank 0's single JSON line and returns the child's exit code.  Under an existing
    launcher (RANK / WORLD_SIZE set) bench.py behaves as before."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
    for ln in child.stdout.splitlines():  # anything else a rank printed goes to stderr: stdout carries ONE line
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if child.returncode == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    return child.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="kitti")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("VH_BENCH_STREAMS", "0")),
                    help="independent camera streams per GPU, stepped together (0: the workload's default)")
    ap.add_argument("--frames", type=int, default=8, help="distinct frames per stream kept in HBM")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--noise", type=int, default=0, help="context runs: +-N grey levels of uniform noise on every frame "
                    "(many more, mostly unmatched features; the driver's line is noise 0, the BASELINE workload)")
    ap.add_argument("--blocks", type=int, default=0, help="timed blocks of --steps steps each (0: as many as make the timed region >= 0.5 s, "
                    "at least 3); every block is bracketed by the barrier + synchronize fence, the median block is reported")
    ap.add_argument("--noise-frac", type=float, default=1.0, help="context runs: share of the pixels that receive --noise")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the as-shipped-loop measurement (e2e_matchfeatures)")
    ap.add_argument("--no-flow", action="store_true", help="skip the reference-pinned mono flow workload (flow_pinned)")
    ap.add_argument("--no-e2e-host", action="store_true", help="skip the host-vote form of the as-shipped loop")
    ap.add_argument("--e2e-steps-per-batch", type=int, default=int(os.environ.get("VH_E2E_STEPS_PER_BATCH", "64")))
    ap.add_argument("--e2e-batches", type=int, default=int(os.environ.get("VH_E2E_BATCHES", "3")))
    ap.add_argument("--e2e-lanes", type=int, default=int(os.environ.get("VH_E2E_LANES", "16")))
    ap.add_argument("--e2e-steps", type=int, default=0, help="steps of the device e2e loop (0: 4 x the steps in flight, at least 48)")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the extra single-stream pass that measures exclusive kernel durations")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="CPU-only rehearsal of the multi-rank plumbing (gloo): sharding, barrier, MAX-reduced timing")
    ap.add_argument("--no-other", action="store_true", help="skip the short context runs of the other BASELINE configs (other_workloads)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    global W, H, NOISE, NOISE_FRAC
    NOISE = args.noise
    NOISE_FRAC = args.noise_frac
    wl = WORKLOADS[args.workload]
    W, H = wl["W"], wl["H"]
    if args.streams <= 0:
        args.streams = wl["streams"]
    if args.dist_selftest:
        return dist_selftest(args)

    # ROCm maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A stream group owns five streams
    # (+ four for the device post stage): with a second group in the process -- the context runs of `other_workloads` beside
    # the idle KITTI group -- streams that should overlap share a queue and the second group runs 16 % slower (1080p: 15.4 k
    # instead of 18.4 k pairs/s; the headline itself: 107.1 -> 107.7 k).  Read when HIP initialises: set before torch is imported.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    # Rank 0 prints ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner at
    # initialisation): the process's fd 1 is pointed at stderr for the run, the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = bool(os.environ.get("VH_BENCH_FORCE_DIST"))  # the env switch rehearses the RCCL plumbing on one GPU
    if world != args.gpus and not force_dist:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...)")
    # RCCL only carries the barrier and the MAX-reduced timing: streams never exchange data
    proto = RankProtocol("nccl", rank, local_rank, world, force=force_dist)
    use_dist = proto.active
    # which device every rank really sits on: (rank, local rank, device index, PCI bus id), gathered to rank 0
    props = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": props.name,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "pci_device_id": getattr(props, "pci_device_id", None),
          "uuid": str(getattr(props, "uuid", ""))}
    ranks_seen = proto.gather(me)
    if rank == 0:
        check_ranks(ranks_seen, world)

    pkg = entry.load_package()
    S, T = args.streams, args.frames
    frames_np, bpl = make_frames(pkg, S, T, rank)
    dims = [W, H, bpl]
    frames = torch.from_numpy(frames_np).to(dev)  # resident in HBM before timing
    stride = H * bpl

    stream = torch.cuda.current_stream()
    grp = pkg.StreamGroup(S, pkg.Params.default(**wl["params"]), device=local_rank, max_features=wl["cap"],
                          max_matches=wl["cap"])
    # every pushBack is ordered after this stream (handle 0, torch's default stream, included): the frames were uploaded on it
    grp.setStream(stream.cuda_stream)

    def step(k):
        t = k % T
        grp.pushBackDevice(frames[t, 0].data_ptr(), frames[t, 1].data_ptr(), stride, dims, False)
        grp.matchFeatures(pkg.METHOD_QUAD)

    def fence():
        grp.synchronize()  # the group's internal detect/match streams
        torch.cuda.synchronize()
        proto.barrier()
        torch.cuda.synchronize()

    k = 0
    for _ in range(args.warmup):
        step(k); k += 1
    fence()

    def one_step():
        nonlocal k
        step(k); k += 1

    # One block of K steps is ~0.1 s at the default K: a single scheduler hiccup would be 5 % of it.  So the
    # block is repeated (every one under the same protocol: EXACTLY --steps steps between two fences, MAX over
    # ranks) until >= 0.5 s have been timed, and the MEDIAN block is the one reported; all block times are in the line.
    block_s = proto.timed_blocks(one_step, fence, args.steps, args.blocks)
    dt = float(np.median(block_s))

    # per-kernel device time: the same loop once more with HIP events recorded around every
    # launch on the stream it is launched on (two internal streams: detect of frame t+1 overlaps
    # match of frame t, so these durations include what the other stream does to the kernel)
    def profiled(group, n, kk, mono=False):
        group.profileReset(); group.profileEnable(True)
        for _ in range(n):
            tq = kk % T
            group.pushBackDevice(frames[tq, 0].data_ptr(), None if mono else frames[tq, 1].data_ptr(), stride, dims, False)
            group.matchFeatures(pkg.METHOD_FLOW if mono else pkg.METHOD_QUAD)
            kk += 1
        group.synchronize(); torch.cuda.synchronize()
        group.profileEnable(False)
        out_ = {}
        for name in KERNELS:
            ms, n_ = group.profileRead(name)
            if n_:
                out_[name] = {"ms_total": ms, "launches": n_, "us_per_launch": 1e3 * ms / n_}
        return out_
    prof = profiled(grp, args.steps, k)
    k += args.steps
    # ... and exclusive durations: a second group whose kernels all run on ONE stream (VH_SERIAL=1)
    prof_excl = {}
    if not args.no_exclusive and rank == 0:
        os.environ["VH_SERIAL"] = "1"
        g2 = pkg.StreamGroup(S, pkg.Params.default(**wl["params"]), device=local_rank, max_features=wl["cap"], max_matches=wl["cap"])
        os.environ.pop("VH_SERIAL")
        g2.setStream(stream.cuda_stream)
        for w_ in range(3):
            g2.pushBackDevice(frames[w_ % T, 0].data_ptr(), frames[w_ % T, 1].data_ptr(), stride, dims, False)
            g2.matchFeatures(pkg.METHOD_QUAD)
        prof_excl = profiled(g2, min(args.steps, 12), 3)
        g2.close()
    # ---- the as-shipped loop (separate keys, never the headline `value`): Matcher::matchFeatures is matching +
    # removeOutliers (src/matcher.cpp:104-108), VisualOdometryStereo::process goes on with bucketFeatures(2, 50, 50) and
    # estimateMotion (src/viso_stereo.cpp:40-51).  `e2e_matchfeatures`: all of it on the GPU (kernels_vote.hip: the
    # Delaunay vote as one lane per match list, several steps in flight); `e2e_matchfeatures_host_vote`: round 3's form,
    # vote + bucketing of step t on the host threads while the GPU computes step t+1.
    e2e = e2e_host = None
    if not args.no_e2e and args.workload == "kitti" and world == 1:  # (like cpu_baseline: a context measurement of the N = 1 run)
        nthr = max(1, min(len(os.sched_getaffinity(0)), 16))
        ego = pkg.EgoParams.default(f=645.24, cu=635.96, cv=194.13, base=0.5707)
        r3 = np.random.default_rng(7).integers(0, 2 ** 31 - 1, (S, ego.ransac_iters, 3)).astype(np.int32)
        cap_ps = int(min(wl["cap"], max(1024, int(nm_max_hint(grp) * 1.25))))
        e2e_metric = "stereo frame-pairs/sec, as-shipped loop: detect + quad match + removeOutliers + bucketFeatures(2,50,50) + stereo estimateMotion"
        dev_gb_before = grp.deviceBytes() / 2**30

        def e2e_device(B, NB, lanes):
            nonlocal k
            depth = B * (NB - 1)
            n_e2e = args.e2e_steps if args.e2e_steps > 0 else max(4 * B * NB, 48)
            grp.postDeviceConfig(B, NB, lanes)
            # untimed: every batch of the ring is used once (their buffers -- tens of GB -- are allocated on first use) and drained
            for j in range(B * NB):
                step(k); k += 1
                grp.postBeginDevice(cap_ps, 2, 50.0, 50.0, ego=ego, rand3=r3)
            for j in range(B * NB):
                grp.postFinishDevice(B * NB - 1 - j)
            fence()
            ok_share, fin_ms, cnts, t_begin, lat_ms = [], [], [], [], []
            t0 = time.perf_counter()
            for j in range(n_e2e + depth):
                if j < n_e2e:
                    t_begin.append(time.perf_counter())  # the step's images are handed over here
                    step(k); k += 1
                    grp.postBeginDevice(cap_ps, 2, 50.0, 50.0, ego=ego, rand3=r3)
                if j >= depth:
                    q = j - depth
                    tq = time.perf_counter()
                    r = grp.postFinishDevice(min(j, n_e2e - 1) - q)
                    fin_ms.append(1e3 * (time.perf_counter() - tq))
                    lat_ms.append(1e3 * (time.perf_counter() - t_begin[q]))  # ... and its poses are in host memory here
                    ok_share.append(float(r["ok"].mean())); cnts.append(float(r["counts"].mean()))
            fence()
            dt_e2e = time.perf_counter() - t0
            return {"metric": e2e_metric, "value": S * n_e2e / dt_e2e, "unit": "pairs/s", "steps": n_e2e, "ms_per_step": 1e3 * dt_e2e / n_e2e,
                   "latency_ms_per_result": {"median": float(np.median(lat_ms)), "max": float(np.max(lat_ms)),
                                             "is": "host clock from handing a step's images over (pushBack) to its S poses being in host memory"},
                   "ring_gb": round(grp.deviceBytes() / 2**30 - dev_gb_before, 1),
                   "where": "device: vote (one lane per match list), bucketing and pose estimate are kernels; the host only begins and finishes steps",
                   "host_ms_per_step_vote_and_bucket": 0.0, "host_ms_per_step_waiting_in_finish": float(np.mean(fin_ms)),
                   "steps_per_batch": B, "batches_in_flight": NB, "lists_per_wave": lanes, "steps_in_flight": depth,
                   "timed_region": "all steps, including the fill and the drain of the pipeline",
                   "bucketed_matches_per_stream": float(np.mean(cnts)), "pose_ok_share": float(np.mean(ok_share)),
                   "slot_records_per_stream": cap_ps}

        # -- device form (context measurements must never cost the line its headline: an error is reported in place of the number)
        B, NB, lanes = args.e2e_steps_per_batch, args.e2e_batches, args.e2e_lanes
        # the ring of batches needs 176 bytes per match slot and list in flight: fewer steps per batch on a device with less free
        # memory (the library does the same on its own: vh_group_post_begin_device sizes the ring against the free memory)
        free_b, _ = torch.cuda.mem_get_info()
        per_step = S * cap_ps * 180.0
        while B > 4 and B * NB * per_step > 0.7 * free_b:
            B //= 2
        try:
            e2e = e2e_device(B, NB, lanes)
            # ... and at operating points a real-time consumer could live with: few steps in flight.  The best rate among
            # them whose median latency stays within 100 ms is reported beside the throughput-optimal point above.
            args_steps_saved, args.e2e_steps = args.e2e_steps, 40
            bounded = []
            for (b_, nb_, l_) in ((1, 3, 1), (1, 5, 4), (2, 4, 8)):
                r_ = e2e_device(b_, nb_, l_)
                bounded.append({k_: r_[k_] for k_ in ("value", "ms_per_step", "latency_ms_per_result", "steps_per_batch", "batches_in_flight", "lists_per_wave", "steps_in_flight")})
            args.e2e_steps = args_steps_saved
            ok_ = [b_ for b_ in bounded if b_["latency_ms_per_result"]["median"] <= 100.0]
            e2e["latency_bounded"] = {"bound_ms": 100.0, "best": max(ok_, key=lambda b_: b_["value"]) if ok_ else None, "tried": bounded,
                                      "note": "one list is one GPU lane for >= 60 ms (the sequential sweep): the device vote cannot deliver within 100 ms; "
                                              "the operating point for a latency-bound consumer is the host-vote form (e2e_matchfeatures_host_vote: its latency key)"}
        except Exception as ex:  # e.g. not enough free HBM for the ring of batches
            e2e = {"error": f"{type(ex).__name__}: {ex}", "steps_per_batch": B, "batches_in_flight": NB, "lists_per_wave": lanes}
            try:
                grp.synchronize()
            except Exception:
                pass
        # -- host-vote form (round 3)
        if not args.no_e2e_host:
            n_h = max(4, min(args.steps, 12))
            host_ms, ok_share, t_beg, lat_h = [], [], [], []
            fence()
            t0 = time.perf_counter()
            for j in range(n_h + 1):
                if j < n_h:
                    t_beg.append(time.perf_counter())
                    step(k); k += 1
                    grp.postBegin(cap_ps)
                if j > 0:
                    r = grp.postFinish(1 if j < n_h else 0, 2, 50.0, 50.0, host_threads=nthr, ego=ego, rand3=r3, want_lists=False)
                    lat_h.append(1e3 * (time.perf_counter() - t_beg[j - 1]))
                    host_ms.append(r["host_ms"]); ok_share.append(float(r["ok"].mean()))
            fence()
            dt_h = time.perf_counter() - t0
            e2e_host = {"metric": e2e_metric, "value": S * n_h / dt_h, "unit": "pairs/s", "steps": n_h, "ms_per_step": 1e3 * dt_h / n_h,
                        "latency_ms_per_result": {"median": float(np.median(lat_h)), "max": float(np.max(lat_h)),
                                                  "is": "host clock from handing a step's images over to its S poses being in host memory (two steps in flight)"},
                        "host_threads": nthr, "host_ms_per_step_vote_and_bucket": float(np.median(host_ms)),
                        "bucketed_matches_per_stream": float(r["counts"].mean()), "pose_ok_share": float(np.mean(ok_share)),
                        "bound": "host: the Delaunay vote of removeOutliers is a sequential float triangulation per stream (csrc/outliers.cpp)"}
    # ---- the reference-pinned workload (a context key): what VisualOdometryMono::process runs (src/viso_mono.cpp:33-39) --
    # pushBack of ONE camera + matchFeatures(0), flow matching being the only composition the reference implements
    # (src/matcher.cpp:308-336).  Same S streams, same frames (left camera); after the timing stream 0's first frame pair,
    # which is SURVEY Appendix B's pair, is matched once more on its own and compared with the reference's known answers.
    flow = None
    if not args.no_flow and args.workload == "kitti" and world == 1:
        gf = pkg.StreamGroup(S, pkg.Params.default(**wl["params"]), device=local_rank, max_features=wl["cap"], max_matches=wl["cap"])
        gf.setStream(stream.cuda_stream)
        kf = 0

        def fstep():
            nonlocal kf
            gf.pushBackDevice(frames[kf % T, 0].data_ptr(), None, stride, dims, False)
            gf.matchFeatures(pkg.METHOD_FLOW)
            kf += 1

        def ffence():
            gf.synchronize(); torch.cuda.synchronize()
        for _ in range(args.warmup):
            fstep()
        ffence()
        fb = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fstep()
            ffence()
            fb.append(time.perf_counter() - t0)
        fdt = float(np.median(fb))
        fprof = profiled(gf, min(args.steps, 12), kf, mono=True)
        kf += min(args.steps, 12)
        fnf, fnm = gf.getCounts()
        fq = [gf.getFeatures(0, w_) for w_ in (0, 2)]  # stream 0: previous / current left features of the last step
        pairs_f = 2 * in_window_pairs(fq[1], fq[0], pkg.Params.default(**wl["params"]).match_radius)  # both passes walk the same in-window pairs
        gf.close()
        fsec = fprof["match"]["us_per_launch"] * 1e-6 if "match" in fprof else None
        known = np.load(os.path.join(ROOT, "tests", "golden", "known_answers.npz"))
        m1 = pkg.Matcher(pkg.Params.default(**wl["params"]), outlier_removal=False)
        m1.pushBack(frames_np[0, 0, 0], None, dims, False)
        m1.pushBack(frames_np[1, 0, 0], None, dims, False)
        m1.matchFeatures(pkg.METHOD_FLOW)
        pm1 = m1.getMatches()
        m1.close()
        flow = {"metric": "mono frames/sec (detect 1 image + flow match), KITTI 1241x376: the composition the reference implements (src/matcher.cpp:308-336)",
                "value": S * args.steps / fdt, "unit": "frames/s", "ms_per_step": 1e3 * fdt / args.steps, "blocks_s": [round(b, 5) for b in fb],
                "features_per_image": float(fnf.astype(np.float64)[:, 2].mean()), "matches_per_frame": float(fnm.mean()),
                "kernels_us_per_launch": {n_: round(v["us_per_launch"], 2) for n_, v in fprof.items()},
                "valu_sad": None if not fsec else {"achieved": pairs_f * 8 / 64 * S / fsec, "peak_measured": 5.26e11, "frac": pairs_f * 8 / 64 * S / fsec / 5.26e11,
                                                   "in_window_pairs_per_frame": pairs_f},
                "known_answer": {"source": "SURVEY Appendix B / tests/golden/known_answers.npz (generated by the reference's own build)",
                                 "matches": int(len(pm1)), "matches_expected": int(known["kitti_1241x376__n_match"]),
                                 "fnv_p_match": "%016x" % pkg.synth.fnv1a64(pm1), "fnv_expected": "%016x" % int(known["kitti_1241x376__fnv_p_match"])}}
        flow["known_answer"]["ok"] = bool(flow["known_answer"]["matches"] == flow["known_answer"]["matches_expected"] and
                                          flow["known_answer"]["fnv_p_match"] == flow["known_answer"]["fnv_expected"])
    search_spec, search_redo = grp.searchStats()
    nf, nm = grp.getCounts()
    wl_radius = pkg.Params.default(**wl["params"]).match_radius
    last = (k - 1) % T
    got0 = grp.getMatches(0)

    # ---- the other BASELINE.json configs and one noisy run, a few steps each (context keys of the N = 1 line, never `value`)
    other = None
    if rank == 0 and world == 1 and args.workload == "kitti" and not args.no_other and not NOISE:
        other = {}
        ob_ = None if args.no_cpu else entry.load_oracle()
        for key, (name_, noise_) in (("1080p", ("1080p", 0)), ("4k", ("4k", 0)), ("kitti_noise1", ("kitti", 1))):
            try:
                other[key] = context_workload(pkg, torch, dev, local_rank, name_, noise_, ob=ob_)
            except Exception as ex:  # a context run must never cost the line its headline
                other[key] = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0:
        pairs = world * S * args.steps
        value = pairs / dt
        # algorithmic bytes per stereo pair: read 2 new images, write 2 new
        # feature sets, the matcher reads all 4 sets once, write M matches
        nfm = nf.astype(np.float64).mean(axis=0)
        B_pair = 2 * bpl * H + 48 * (nfm[2] + nfm[3]) + 48 * nfm.sum() + 48 * float(nm.mean())
        dom = max(prof, key=lambda n_: prof[n_]["ms_total"]) if prof else None
        roofline = None
        if dom:
            sec = prof[dom]["us_per_launch"] * 1e-6
            achieved = S * B_pair / sec / 1e9
            # HBM bytes of that kernel per launch from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes
            # (tools/profile_round.sh -> tools/summarize_profiles.py): reported when they were taken on THIS code
            # (sha256 of the sources that determine the code object -- the .so itself hashes differently on every
            # rebuild) with this number of streams; traffic_step = sum over the step's kernels x launches
            traffic, traffic_src, traffic_step = None, None, None
            tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tj):
                try:
                    tr = json.load(open(tj))
                    src_sha = pkg.source_sha256()
                    if tr.get("streams") == S and tr.get("kernel") == dom and tr.get("src_sha256") == src_sha:
                        traffic = tr.get("hbm_bytes_per_launch")
                        traffic_src = f"profiles/traffic_latest.json ({tr.get('tag')}, sources {src_sha[:12]})"
                        lps = {n_: v["launches"] / args.steps for n_, v in prof.items()}
                        raw = sum(tr["per_kernel_raw"].get(n_, 0.0) * l_ for n_, l_ in lps.items())
                        cor = sum(tr["per_kernel"].get(n_, 0.0) * l_ for n_, l_ in lps.items())
                        traffic_step = {"unit": "bytes per step of S pairs", "raw": raw, "corrected_fetch_x2": cor,
                                        "algorithmic": S * B_pair, "over_algorithmic_raw": raw / (S * B_pair),
                                        "over_algorithmic_corrected": cor / (S * B_pair),
                                        "hbm_frac_of_peak_corrected": cor / dt * args.steps / 1e9 / HBM_PEAK_GBS}
                    else:
                        traffic_src = "profiles/traffic_latest.json is from other sources or another stream count: not reported"
                except Exception as ex:
                    traffic, traffic_src = None, f"profiles/traffic_latest.json unreadable: {type(ex).__name__}"
            # What binds: the searches' compulsory v_sad_u8 work against the issue rate of that
            # instruction (tools/ubench_valu.hip, profiles/r02_ubench_valu.txt: 4.27 SIMD cycles per
            # wave-instruction at >= 2 waves per SIMD, i.e. half rate, 5.26e11 wave-instructions/s over
            # the 1024 SIMDs at the 2.2 GHz the chip holds under it) -- 8 SADs per in-window pair
            valu = None
            if dom == "match" and nfm.mean() < 20000:  # (the pair count below is O(N^2) host work)
                pairs_ = []
                for s_ in range(min(S, 3)):
                    f = [grp.getFeatures(s_, w_) for w_ in range(4)]
                    pairs_.append(in_window_pairs(f[1], f[3], wl_radius) + in_window_pairs(f[2], f[0], wl_radius))
                sad_wave_instr = float(np.mean(pairs_)) * 8 / 64 * S
                peak = 5.26e11
                valu = {"unit": "v_sad_u8 wave-instructions/s", "achieved": sad_wave_instr / sec, "peak_measured": peak,
                        "frac": sad_wave_instr / sec / peak, "in_window_pairs_per_stereo_pair": float(np.mean(pairs_)),
                        "note": "compulsory SADs of the two flow passes only (the speculative search executes ~1.4x as many; "
                                "pairs outside a query's window count as overhead, not as work)"}
                if dom in prof_excl:
                    valu["frac_exclusive"] = sad_wave_instr / (prof_excl[dom]["us_per_launch"] * 1e-6) / peak
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_step": traffic_step,
                        "algorithmic_bytes_per_launch": S * B_pair, "us_per_launch": prof[dom]["us_per_launch"],
                        "us_per_launch_is": "overlapped with the other internal stream (as in the timed region)",
                        "binding_resource": "VALU issue of v_sad_u8 (half-rate instruction), not HBM",
                        "valu_sad": valu,
                        "note": "integer SAD search: v_sad_u8 issue-bound, not HBM-bound (DESIGN.md section 4)"}
            if dom in prof_excl:
                ex = prof_excl[dom]["us_per_launch"] * 1e-6
                roofline["exclusive"] = {"us_per_launch": prof_excl[dom]["us_per_launch"], "achieved": S * B_pair / ex / 1e9,
                                         "frac": S * B_pair / ex / 1e9 / HBM_PEAK_GBS,
                                         "note": "the same kernel with every kernel of the step on one stream (VH_SERIAL=1)"}
        out = {
            "metric": METRIC if args.workload == "kitti" else f"stereo frame-pairs/sec (detect+match), {W}x{H}; matches bit-exact",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "blocks": {"count": len(block_s), "steps_each": args.steps, "reported": "median", "seconds_each": [round(b, 5) for b in block_s],
                       "timed_seconds_total": round(float(np.sum(block_s)), 4)},
            "ranks_seen": ranks_seen,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["label"] + (f" + noise +-{NOISE}" if NOISE else ""), "noise": NOISE,
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "streams_per_gpu": S, "frames_in_hbm": T, "features_per_image": float(nfm.mean()),
                       "matches_per_pair": float(nm.mean()), "parallelism": f"{world}x independent stream groups",
                       "device_mib_per_stream": round(grp.deviceBytes() / S / 2**20, 2)},
            "roofline": roofline,
            "search_loop": {"form": "speculative" if search_spec else "tested", "queries_searched_again": search_redo,
                            "note": "chosen per launch from the share of queries whose best candidate over the walked region "
                                    "lies outside their own window (DESIGN.md 4.1); results are identical either way"},
            "kernels_us_per_launch": {n_: round(v["us_per_launch"], 2) for n_, v in prof.items()},
            # detection runs in sub-batches of streams (engine.hip): several launches per step
            "kernels_launches_per_step": {n_: round(v["launches"] / args.steps, 2) for n_, v in prof.items()},
            "kernels_us_per_launch_exclusive": {n_: round(v["us_per_launch"], 2) for n_, v in prof_excl.items()},
            "flow_pinned": flow,
            "other_workloads": other,
            "e2e_matchfeatures": e2e,
            "e2e_matchfeatures_host_vote": e2e_host,
            "parity_scope": "primitives (computeFeatures, createIndexVector, findMatch, flow matching) pinned to the reference; "
                            "stereo/quad composition per SURVEY A.7 (absent from the reference: unpinned)",
        }
        if not args.no_cpu:
            ob = entry.load_oracle()
            rate, n_pairs, secs, results = cpu_baseline(ob, frames_np, dims, args.cpu_seconds, wl["params"])
            calib = None
            try:
                calib = json.load(open(os.path.join(ROOT, "profiles", "r02_cpu_calibration.json")))["port_over_reference_sse"]
            except Exception:
                pass
            out["cpu_baseline"] = {"value": rate, "unit": "pairs/s", "cores": 1, "kind": "port",
                                   "port_over_reference_sse": calib,
                                   "port_over_reference_sse_source": "profiles/r02_cpu_calibration.json (tools/calibrate_cpu_baseline.py: the "
                                                                      "reference's SSE build and this port on one core of the build container, cfg-1)",
                                   "sample": f"{n_pairs} consecutive stereo pairs of stream 0 (detect 2 images + quad match each), "
                                             f"{secs:.1f} s, oracle/viso_oracle.c single thread"}
            nthr = max(1, min(len(os.sched_getaffinity(0)), S, 16))  # 16 = the host-core share of one GPU on this pool
            if nthr > 1:
                mrate, mpairs, msecs = cpu_baseline_threads(ob, frames_np, dims, min(args.cpu_seconds, 6.0), wl["params"], nthr)
                out["cpu_baseline_all_cores"] = {"value": mrate, "unit": "pairs/s", "cores": nthr, "kind": "port",
                                                 "sample": f"{mpairs} stereo pairs over {nthr} streams, one host thread each, {msecs:.1f} s"}
            # the oracle as checker: stream 0's last GPU step must equal the CPU result for the same frames
            o = ob.Oracle(); p = ob.Params.default(**wl["params"])
            prev_t = (last - 1) % T
            f = [o.compute_features(p, frames_np[t_, c, 0], dims)[1] for t_ in (prev_t, last) for c in (0, 1)]
            want = o.matching(p, dims, 2, *f)
            out["parity_checked"] = bool(got0.tobytes() == want.tobytes())
        print(json.dumps(out), file=real_stdout, flush=True)
    grp.close()
    proto.close()


if __name__ == "__main__":
    sys.exit(main() or 0)
