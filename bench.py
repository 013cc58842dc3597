#!/usr/bin/env python
"""bench.py -- MC trial moves/s of the per-move energy hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W [--replicas R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]/[2]): SPC/E water, 750 molecules (NIST sample configuration 4 ==
Ewald/coord750.txt, shipped as metropolismontecarlo_amd/data/spce_nist.npz), NVT at 298.15 K, full Ewald
(kappa = 5.6/L, 337 k-vectors), fp64, r_cut = 10 A.  R independent replicas per GPU (one Markov
chain each; chain r of rank k draws from the stream (seed, k*R + r)); replicas shard across ranks
with no data-path collective (weak scaling: R per GPU is fixed).  A *step* is one trial move of
every replica of the rank: the fused move kernel (2x LJ_poly_dU + 2x EwaldShort + RecipMove +
commit of the previous accepted move) and the sequential Metropolis accept/reject of
Loop() (main.jl:593-651) for every replica.  By default the kernel takes the decision itself (same
Philox uniform, same arithmetic: the chains are the host-decided ones bit for bit) and one launch
per replica group takes every replica through eight consecutive steps, sending one result record
per replica and launch to the host, which keeps the books (native C++ driver, mmc_batch_run);
`--accept host` leaves the decision to the host's threads, one launch and 64 B per step.  Inputs are
resident in HBM before the timed region; the trial moves are drawn on the device (k_propose,
Philox4x32-10; --device-moves 0: the host draws the moves and sends 232 B per replica and step).

The default R = 61440 fills the device (the headline line): two groups of 30720 = six replicas for
each of the 5120 wavefronts the move kernel keeps resident, so that the waves of a launch end together.  BASELINE's two named replica counts
are first-class too: `--replicas 1` (configs[1]) and `--replicas 32` (configs[2]'s share of one
GPU) print the same contract line with their own roofline object; the default run also carries
both as `named_configs`.

One JSON line on stdout (rank 0).  Extra objects:
  roofline       dominant kernel (k_move_eval_wave for launches of >= 16 moves per CU, else
                 k_move_eval_fast): algorithmic bytes per launch (78.7 KB per
                 trial move at 750 molecules x moves per launch) / average launch duration, against
                 the 8 TB/s HBM peak -- the contract's figure, NOT this kernel's ceiling -- and
                 `binding`: the resource that does bind (fp64 vector issue; HBM on counter bytes),
                 each with a frac <= 1 computed from this run's timing (see roofline_object for
                 how a launch is timed when launches overlap).  `traffic`, instructions per move
                 and the clock are NOT measured in this run (PMC counters cannot be read from
                 inside the process): they are the rocprofv3 figures of the same command committed
                 under profiles/, quoted only when the launch shape matches (`*_source`).
  full_energy_eval  M2: ns per potential(..., "ewald"), batched over the replicas and as the
                 latency of ONE system (750 and 10 000 molecules), each with its fraction of the
                 fp64 vector peak on the survey's flop count.
  call_surface   the reference's OWN call surface on one chain: microseconds per LJ_poly_dU /
                 EwaldShort / RecipMove call and per Loop() body (Ewald/main.jl:487-644) written with
                 those calls through metropolismontecarlo_amd/api.py (one library call each; what
                 MMCHip.jl's ccalls cost plus Python), per move through mmc_trial_move, and the same
                 with a kernel launch per evaluation instead of the context's persistent kernel.
  cpu_baseline   the CPU oracle (a single-threaded C port of the reference's Julia code path; the
                 Julia reference itself cannot run here) timed on this host on the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TEMPERATURE = 298.15      # Ewald/main.jl:62
DR_MAX = 0.316555789      # Ewald/main.jl:118
DPHI_MAX = 0.05           # Ewald/main.jl:73
RCUT = 10.0               # Ewald/main.jl:67
SEED = 11234              # Monatomic/mainMonatomic.jl:15
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6   # fp64 vector peak (datasheet)
N_K = 337
N_CUS = 256
N_SIMDS = 4 * N_CUS


def algorithmic_bytes_per_move(n_mol, box, r_cut=RCUT, n_k=N_K):
    """B_move = 2*(24*N_mol + 108*Mbar) + 52*N_k with
    Mbar = 4/3 pi r_cut^3 * N_mol / L^3."""
    mbar = 4.0 / 3.0 * np.pi * r_cut ** 3 * n_mol / box ** 3
    return 2 * (24 * n_mol + 108 * mbar) + 52 * n_k


def algorithmic_bytes_full_eval(n_mol, n_k=N_K):
    """B_full = 36*N + 24*N_mol + 36*N_k."""
    return 36 * 3 * n_mol + 24 * n_mol + 36 * n_k


def algorithmic_flops_full_eval(n_mol, box, r_cut=RCUT, n_k=N_K):
    """half-pair COM tests (12 flop), 9 x 48-flop atom-pair terms + one
    12-flop LJ term per molecule pair inside the gate, 16-flop phase terms, 120 flop of phase
    set-up per atom."""
    mbar = 4.0 / 3.0 * np.pi * r_cut ** 3 * n_mol / box ** 3
    return (12 * n_mol * (n_mol - 1) / 2 + (9 * 48 + 12) * n_mol * mbar / 2
            + 16 * n_k * 3 * n_mol + 120 * 3 * n_mol)


def pmc_traffic(kernel, moves_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE x2 on gfx950; profiles/*_traffic.json),
    only when this run has the kernel and launch shape the counters were collected for."""
    try:
        path = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles"))
                      if p.endswith("_traffic.json"))[-1]
        t = json.load(open(os.path.join(ROOT, "profiles", path)))
    except (IndexError, OSError, ValueError):
        return None, None
    if t.get("kernel") != kernel or int(moves_per_launch) != int(t["moves_per_launch"]):
        return None, None
    return t["bytes_per_launch"], (f"profiles/{path}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                   "command (committed); not measured in this run")


def access_pattern_floor(units_per_launch, steps_per_launch):
    """(microseconds per launch, TB/s of the pattern's bytes, source) of scripts/gather_bw.hip: the move
    kernel's memory accesses -- per unit and step the COM-code stream, 125 scattered 128-byte record
    lines, S(k) read and written; same launch shape, the same number of consecutive steps per unit by
    the same wave -- replayed WITHOUT its arithmetic (committed under profiles/; not measured in
    this run).  None unless the replay was made for this launch shape."""
    try:
        path = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles"))
                      if p.endswith("_access_pattern_bw.json"))[-1]
        t = json.load(open(os.path.join(ROOT, "profiles", path)))
        if int(t["units_per_launch"]) != int(units_per_launch):
            return None
        if int(steps_per_launch) <= 1:
            c = next(c for c in t["cases"] if c["pattern"].startswith("the move kernel's mix"))
            us, tbs = c["us_per_launch"], c["TB_per_s"]
        else:
            c = next(c for c in t["steps_per_unit"] if int(c["K"]) == int(steps_per_launch))
            us, tbs = c["us_per_unit_steps_of_one_launch"] * steps_per_launch, c["TB_per_s_of_the_pattern_bytes"]
        return us, tbs, (f"profiles/{path}: scripts/gather_bw.hip, the kernel's loads and stores without its "
                         "arithmetic, same units and steps per launch (committed; not measured in this run)")
    except (IndexError, OSError, ValueError, KeyError, StopIteration):
        return None


def pmc_extras(kernel, moves_per_launch):
    """What the committed rocprofv3 PMC passes of this command say beyond bytes (profiles/
    *_default_pmc_summary.json, written by scripts/summarize_profile.py): VALU instructions per
    move, the share of cycles the vector pipe issues, the clock.  Only for the kernel and launch
    shape they were collected for; not measured in this run."""
    try:
        path = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles"))
                      if p.endswith("_default_pmc_summary.json"))[-1]
        t = json.load(open(os.path.join(ROOT, "profiles", path)))
        ex = t["roofline_extras"]
    except (IndexError, OSError, ValueError, KeyError):
        return {}
    if ex.get("kernel") != kernel or int(ex.get("moves_per_launch", -1)) != int(moves_per_launch):
        return {}
    out = {k: ex[k] for k in ("valu_insts_per_move", "valu_busy_frac", "clock_ghz", "hbm_traffic_gbs",
                              "hbm_traffic_frac", "waves_per_simd", "lds_busy_frac", "lds_conflict_frac",
                              "wait_frac", "salu_insts_per_move") if k in ex}
    out["counters_source"] = f"profiles/{path} (rocprofv3 --pmc passes of this command; not measured in this run)"
    return out


def cpu_baseline(a, budget_s, n_threads=1):
    """Time the oracle (C port of the reference path) on the same workload: Loop()'s hot-path
    calls for successive molecules with small rigid translations, in a C loop
    (orc_bench_trial_moves; n_threads independent chains, one per thread) for `budget_s` seconds.
    Returns (moves/s, moves, seconds, seconds of one full energy evaluation on one thread)."""
    from oracle import oracle as orc
    s = orc.System(a["com"], a["first_atom"], a["last_atom"], a["coords"], a["atype"], a["charge"],
                   a["eps"], a["sig"], a["box"])
    ew = orc.Ewald(5.6 / s.box, 5, 27, s.box)
    orc.recip_long(ew, s.coords, s.charge, s.box)
    n, dt = orc.bench_trial_moves(s, ew, RCUT, RCUT, DR_MAX, SEED, n_threads, budget_s)
    t1 = time.perf_counter()
    orc.potential_ewald(s, ew, RCUT, RCUT)
    t_full = time.perf_counter() - t1
    return n / dt, n, dt, t_full


def default_parts(R, n_mol):
    """The library's default number of units per move (batch_default_parts, csrc/mmc_batch.inc)."""
    if R >= 8192:
        return 1
    return int(max(1, min((2048 + R - 1) // R, 1 + (n_mol + 187) // 188, 16)))


def kernel_name(kernel_opt, moves_per_launch, parts):
    if kernel_opt == 0:
        return "k_move_eval"
    if kernel_opt in (1, 2):
        return {1: "k_move_eval_fast", 2: "k_move_eval_wave"}[kernel_opt]
    return "k_move_eval_wave" if moves_per_launch * parts >= 16 * N_CUS else "k_move_eval_fast"


def server_lat_parts(R, n_mol):
    """Parts (waves) of the latency move server the library takes for R replicas (batch_lat_shape,
    csrc/mmc_batch.inc), or 0 when it takes the one-workgroup server or none."""
    if R > N_CUS:
        return 0
    def applies(G):
        P = 4 * G
        nr = 3 if P >= 12 else 2
        return -(-n_mol // (P - nr)) <= 128
    G = 4 if 4 * R <= N_CUS else 2
    while G >= 2:
        if applies(G):
            return 4 * G
        G -= 1
    G = (-(-n_mol // 128) + 3 + 3) // 4     # a large system: as many workgroups as it takes
    while G <= 32 and not applies(G):
        G += 1
    return 4 * G if G <= 32 and R * G <= N_CUS else 0


def shape_for(R, args):
    """Launch shape for R replicas per GPU: groups, host threads, steps, warm-up.  Small batches
    are latency-bound: one group for a single chain, proposals read in place (no H2D copy in the
    step), and enough steps for a stable clock."""
    small = R < 4096
    groups = args.groups if args.groups > 0 else (1 if R == 1 else 2)
    threads = args.threads
    if small: # the move server steps every replica at its own pace: a thread per ~8 replicas
        threads = max(1, min(threads, R // 8))
    steps = args.steps if args.steps is not None else (3000 if small else 600)
    warmup = args.warmup if args.warmup is not None else (300 if small else 64)  # (multiples of the 8 steps a launch takes)
    zero_copy = args.zero_copy_moves if args.zero_copy_moves >= 0 else (1 if small else 0)
    prewarm = max(0, (int(os.environ.get("MMC_BENCH_PREWARM", "120")) if not small else 600) - warmup)
    return dict(groups=groups, threads=max(threads, 1), steps=steps, warmup=warmup,
                zero_copy=zero_copy, prewarm=prewarm)


def measure_moves(R, a, args, local_rank, g0, barrier, shape, n_parts=None, persistent=None):
    """One timed run of the native driver on a fresh batch of R replicas.  Returns the figures of
    this rank; the caller reduces over ranks.  `persistent`: the move server for small batches
    (None = the --persistent flag; the library's default takes it up to one replica per compute unit)."""
    from metropolismontecarlo_amd import sharding, structs
    from metropolismontecarlo_amd.device import Batch
    box = a["box"]
    b = Batch(R, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], box,
              5.6 / box, structs.factor, RCUT, RCUT, device=local_rank)
    b.set_option("kernel", args.kernel)
    if args.wave_wgs:
        b.set_option("wave_wgs", args.wave_wgs)
    b.set_option("zero_copy_moves", shape["zero_copy"])
    b.set_option("device_moves", args.device_moves)
    b.set_option("accept_on_device", {"auto": -1, "host": 0, "kernel": 1}[args.accept])
    if (args.persistent if persistent is None else persistent) != -1:
        b.set_option("persistent", args.persistent if persistent is None else persistent)
    parts = args.parts if n_parts is None else n_parts
    # initial total energy of every replica (also initialises S(k)): the M2 metric, batched
    b.potential_ewald(as_array=True)
    barrier()
    t0 = time.perf_counter()
    tot = b.potential_ewald(as_array=True)      # mmc_totals[R] written straight into numpy
    t_full = time.perf_counter() - t0
    energies = tot["energy"].copy()
    # every Nth launch of a group is bracketed by events; short runs sample more densely
    ev = 0 if args.no_events else max(1, min(args.event_every, shape["steps"] // 4))
    kw = dict(n_groups=shape["groups"], n_parts=parts, time_kernels=ev, n_threads=shape["threads"],
              n_streams=args.streams, replica0=g0)
    # Steady state first: the move kernel's launch time keeps falling over the first ~50 launches
    # of a process (377 -> 355 -> 343 us in three successive 20-step calls), so a short --warmup
    # would time a device that is still warming up.  These steps are untimed, like the --warmup
    # ones that follow, and reported as config.prewarm_steps.
    if shape.get("prewarm", 0) > 0:
        energies, _ = b.run(shape["prewarm"], TEMPERATURE, DR_MAX, DPHI_MAX,
                            sharding.run_seed(phase=2), energies, **kw)
    energies, _ = b.run(shape["warmup"], TEMPERATURE, DR_MAX, DPHI_MAX, sharding.run_seed(phase=0),
                        energies, **kw)
    barrier()
    t0 = time.perf_counter()
    energies, st = b.run(shape["steps"], TEMPERATURE, DR_MAX, DPHI_MAX, sharding.run_seed(phase=1),
                         energies, **kw)
    barrier()
    elapsed = time.perf_counter() - t0
    # consistency: running totals vs a full recompute (Poly/main.jl:232-235), outside the timing
    tot2 = b.potential_ewald(as_array=True)
    drift = float(np.max(np.abs(energies - tot2["energy"]) / np.abs(energies)))
    b.close()
    streams = args.streams if args.streams > 0 else (min(shape["groups"], 2) if args.device_moves else shape["groups"])
    return dict(st=st, elapsed=elapsed, t_full=t_full, drift=drift, energy_sum=float(energies.sum()),
                launches_per_step=st["launches"] / max(shape["steps"], 1),
                server=st["server_steps"] > 0, streams=streams)  # (st["device_decisions"]: who decided)


def launch_mode_roofline(R, a, args, local_rank, g0, barrier, shape, n_mol, box, parts_used):
    """The move server has no launches to time: the roofline object of a small batch comes from a
    second, short run of the same batch with a launch per step (persistent = 0)."""
    sh = dict(shape, steps=min(shape["steps"], 600), warmup=min(shape["warmup"], 60))
    lat_parts = server_lat_parts(R, n_mol)
    if lat_parts and args.kernel == 3:
        # the server ran k_move_server_lat: its launch-per-step form is k_move_eval_lat with the
        # same parts (same arithmetic, bit-identical chains -- tests/test_gpu_server.py)
        a2 = argparse.Namespace(**{**vars(args), "kernel": 4})
        r_ = measure_moves(R, a, a2, local_rank, g0, barrier, sh, n_parts=lat_parts, persistent=0)
        rf = roofline_object(r_, R, a2, sh, n_mol, box, lat_parts // 4)
        if rf:
            rf["kernel"] = "k_move_eval_lat"
    else:
        r_ = measure_moves(R, a, args, local_rank, g0, barrier, sh, persistent=0)
        rf = roofline_object(r_, R, args, sh, n_mol, box, parts_used)
    if rf:
        rf["measured_in"] = ("a separate run with a launch per step (persistent = 0) of the kernel "
                             "the move server is the persistent form of")
        rf["us_per_step_launch_per_step"] = 1e6 * r_["elapsed"] / sh["steps"]
    return rf


def roofline_object(res, R, args, shape, n_mol, box, parts_used):
    """The contract's roofline object for the dominant kernel, from THIS run's timing.

    Launch duration.  One stream: the HIP-event duration of a launch (events on the kernel's own
    stream, every --event-every'th launch of the timed region) -- what rocprofv3's kernel trace
    shows.  Two streams (the default with device-side proposals): launches of the two replica
    groups overlap, an event pair then measures a launch's SPAN while it shares the GPU
    (`launch_span_us`, also what rocprofv3 shows), not what it costs; the cost of a launch is the
    timed region's wall time -- during which the GPU always has a move kernel running -- divided
    by its launches (`avg_launch_us`; fill and drain of the pipeline included, so it errs high).

    `achieved` / `frac` are the contract's ALGORITHMIC bytes over that duration and are
    not a ceiling for this kernel (it moves 0.42 of them); what binds is in `binding`: fp64 vector
    issue -- vector instructions per move x moves per launch / duration against 1024 SIMDs x clock
    / 4 cycles per wave64 instruction -- and HBM throughput on counter bytes.  Instructions per
    move, bytes per move and the clock under this load come from the committed rocprofv3 PMC
    passes of this command (profiles/, `counters_source`), the durations from this run."""
    st = res["st"]
    if not st["launches"]:
        return None
    bytes_move = algorithmic_bytes_per_move(n_mol, box)
    moves_per_launch = st["moves"] / max(st["launches"], 1)
    units_per_launch = max(R, 1) / max(shape["groups"], 1)            # replicas a launch takes
    avg_steps = moves_per_launch / units_per_launch                   # e.g. 6.67 for launches of 8 + 8 + 4 steps
    steps_per_launch = next((k for k in (1, 2, 4, 8, 16) if k >= avg_steps - 1e-9), 16)   # the launches' nominal length
    profile_shape = units_per_launch * steps_per_launch               # the launch the committed counters belong to
    overlapped = res.get("streams", 1) > 1 and shape["groups"] > 1
    span = (st["kernel_ms"] * 1e-3 / st["timed_launches"]) if st["timed_launches"] else None
    if overlapped or span is None:
        t_launch = res["elapsed"] / st["launches"]
        t_source = ("timed region wall time / launches: launches of the replica groups overlap on two "
                    "streams, the GPU runs a move kernel throughout")
    else:
        t_launch = span
        t_source = "HIP events around single launches (every Nth) on the kernel's stream"
    achieved = bytes_move * moves_per_launch / t_launch / 1e9
    name = kernel_name(args.kernel, moves_per_launch, parts_used)
    # (the committed counters are per launch of the nominal length: per move they hold for a call
    # whose last launch is shorter, too)
    traffic, source = pmc_traffic(name, profile_shape)
    if traffic:
        traffic *= moves_per_launch / profile_shape
    ex = pmc_extras(name, profile_shape)
    out = {
        "kernel": name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
        "frac_note": ("frac is the contract's ALGORITHMIC bytes (SURVEY 8d: COM scan and neighbour gather "
                      "counted once per state, 24 B per centre of mass) over time; the kernel moves 0.42 of "
                      "them (one scan and one gather for both states, 6-byte codes), so this frac may exceed "
                      "1 and is not the ceiling: see `binding`") if name == "k_move_eval_wave" else
                     ("latency: a launch of this size is a chain of dependent latencies, not a stream of "
                      "bytes -- see DESIGN.md section 4"),
        "avg_launch_us": 1e6 * t_launch, "avg_launch_us_is": t_source,
        "launch_span_us": 1e6 * span if span is not None else None,
        "launches": int(st["launches"]),
        "launches_timed_with_events": int(st["timed_launches"]),
        "algorithmic_bytes_per_move": bytes_move, "moves_per_launch": moves_per_launch,
        "steps_per_launch": steps_per_launch,
        "launches_per_move": st["launches"] / max(st["moves"], 1),
        "frac_of_measured_copy_peak_6290": achieved / 6290.0,
    }
    if ex.get("valu_insts_per_move") and ex.get("clock_ghz"):
        insts = ex["valu_insts_per_move"] * moves_per_launch / t_launch / 1e9   # G wave-instructions / s
        peak = N_SIMDS * ex["clock_ghz"] / 4.0
        valu = {"achieved": insts, "peak": peak, "unit": "G wave64 instructions/s", "frac": insts / peak,
                "valu_insts_per_move": ex["valu_insts_per_move"], "clock_ghz": ex["clock_ghz"],
                "peak_is": f"{N_SIMDS} SIMDs x clock / 4 cycles per wave64 vector instruction (fp64 and fp32 alike)"}
        b = {"bound": "fp64_valu_issue", **{k: valu[k] for k in ("achieved", "peak", "unit", "frac")},
             "fp64_valu_issue": valu, "counters_source": ex["counters_source"]}
        if traffic:
            hb = traffic / t_launch / 1e9
            b["hbm"] = {"achieved": hb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hb / HBM_PEAK_GBS,
                        "bytes_per_move": traffic / moves_per_launch,
                        "frac_of_measured_copy_peak_6290": hb / 6290.0}
        # what the kernel's own loads and stores take with the arithmetic removed: the time of that
        # replay over the time of the launch (<= 1: how close the kernel is to being nothing but its
        # memory accesses)
        pat = access_pattern_floor(units_per_launch, steps_per_launch)
        if pat:
            pat = (pat[0] * avg_steps / steps_per_launch, pat[1], pat[2])   # this run's average launch
            pf = pat[0] * 1e-6 / t_launch
            b["access_pattern"] = {"floor_us": pat[0], "frac": pf, "TB_per_s_of_its_bytes_without_arithmetic": pat[1],
                                   "source": pat[2]}
            if pf > valu["frac"]:   # the larger fraction names the bound
                b.update(bound="hbm_access_pattern", achieved=1e6 * t_launch, peak=pat[0], unit="us per launch (floor / achieved)",
                         frac=pf, peak_is="the time the kernel's own loads and stores take with the arithmetic "
                                          "removed (scripts/gather_bw.hip: code stream, scattered 128-byte record "
                                          "lines, S(k) read and written; same launch shape and steps per launch) "
                                          "over the time of the launch")
        for k in ("lds_busy_frac", "lds_conflict_frac", "wait_frac", "salu_insts_per_move", "waves_per_simd"):
            if k in ex:
                b[k] = ex[k]
        out["binding"] = b
    return out


def single_system_latency(a, local_rank):
    """Latency of ONE potential(..., "ewald") call (what a Julia caller of `potential` sees), at
    750 molecules (NIST configuration 4) and at 10 000 (the lattice of BASELINE configs[3])."""
    from metropolismontecarlo_amd import io as mio
    from metropolismontecarlo_amd import structs
    from metropolismontecarlo_amd.device import Context
    out = {}
    for nm in (750, 10000):
        if nm == 750:
            s, first, last = a, a["first_atom"], a["last_atom"]
        else:
            box4, com4, coords4 = mio.cubic_lattice_water(nm, 0.033101144, "spce", seed=SEED)
            first = 3 * np.arange(nm, dtype=np.int64) + 1
            last = first + 2
            s = dict(com=com4, coords=coords4, atype=np.tile([1, 2, 2], nm),
                     charge=np.tile([mio.SPCE_Q_O, mio.SPCE_Q_H, mio.SPCE_Q_H], nm), eps=a["eps"],
                     sig=a["sig"], box=box4)
        ctx = Context(local_rank)
        ctx.upload_system(s["com"], first, last, s["coords"], s["atype"], s["charge"], s["eps"],
                          s["sig"], s["box"])
        ctx.prepare_ewald(5.6 / s["box"], 5, 27, s["box"], structs.factor)
        ctx.potential_ewald(RCUT, RCUT)
        n = 50 if nm == 750 else 20
        t0 = time.perf_counter()
        for _ in range(n):
            e = ctx.potential_ewald(RCUT, RCUT)["energy"]
        dt = (time.perf_counter() - t0) / n
        fl = algorithmic_flops_full_eval(nm, s["box"])
        out[f"{nm}_molecules"] = {"us": 1e6 * dt, "energy_K": e, "algorithmic_flops": fl,
                                  "frac_fp64_vector_peak": fl / dt / 1e12 / FP64_PEAK_TFLOPS}
        if nm == 10000:   # configs[3]: NPT volume move = snapshot + K6 + table + K2 + K3, volume +-0.5 %
            # full moves through mmc_volume_trial / accept / reject (nothing crosses PCIe but the
            # totals): the ACCEPTED leg alternates between two volumes, the REJECTED leg tries the
            # larger volume and goes back every time
            n_vol = 20
            L0 = s["box"]
            L1 = (L0 ** 3 * 1.005) ** (1.0 / 3.0)
            ctx.volume_trial(L1, 5.6 / L1, RCUT, RCUT)
            ctx.volume_reject()
            t0 = time.perf_counter()
            for i in range(n_vol):
                L4 = L1 if i % 2 == 0 else L0
                e4 = ctx.volume_trial(L4, 5.6 / L4, RCUT, RCUT)["energy"]
                ctx.volume_accept()
            t_acc = (time.perf_counter() - t0) / n_vol
            t0 = time.perf_counter()
            for i in range(n_vol):
                ctx.volume_trial(L1, 5.6 / L1, RCUT, RCUT)
                ctx.volume_reject()
            t_rej = (time.perf_counter() - t0) / n_vol
            e_back = ctx.potential_ewald(RCUT, RCUT)["energy"]
            out["npt_volume_move_10000"] = {"ms": 1e3 * 0.5 * (t_acc + t_rej), "ms_accepted": 1e3 * t_acc,
                                            "ms_rejected": 1e3 * t_rej, "energy_K": e4,
                                            "energy_after_rejections_K": e_back,
                                            "host_round_trips": 0}
        ctx.close()
        if nm == 10000:   # configs[3] as a CHAIN: sweeps of trial moves and volume moves on ONE device state
            from metropolismontecarlo_amd.device import Batch
            b = Batch(1, s["com"], s["coords"], s["atype"], s["charge"], s["eps"], s["sig"], s["box"],
                      5.6 / s["box"], structs.factor, RCUT, RCUT, device=local_rank)
            b.set_option("device_moves", 1)
            e0 = float(b.potential_ewald(as_array=True)["energy"][0])
            vmax = 0.002 * s["box"] ** 3          # dV uniform in +-0.1 % of V
            e0, _, _ = b.run_npt(1, TEMPERATURE, 0.0, vmax, DR_MAX, DPHI_MAX, SEED, e0, moves_per_sweep=300)
            n_sw = 3
            t0 = time.perf_counter()
            e1, st, ns = b.run_npt(n_sw, TEMPERATURE, 0.0, vmax, DR_MAX, DPHI_MAX, SEED + 1, e0)
            dt = time.perf_counter() - t0
            e2 = float(b.potential_ewald(as_array=True)["energy"][0])
            out["npt_sweep_10000"] = {
                "workload": "BASELINE configs[3]: ONE chain of 10 000 SPC/E molecules, NPT: sweeps of 10 000 trial "
                            "moves (Loop(), main.jl:487-644) each followed by one volume move "
                            "(volumeChange.jl:59-147: rescale, k-vectors and tables rebuilt, total energy), "
                            "all on one device state (mmc_batch_run_npt)",
                "sweeps": n_sw, "trial_moves": int(st["moves"]), "volume_moves": int(ns["vol_attempt"]),
                "moves_per_s_including_volume_moves": (st["moves"] + ns["vol_attempt"]) / dt,
                "us_per_trial_move": 1e3 * st["wall_ms"] / max(st["moves"], 1),
                "ms_per_volume_move": ns["volume_ms"] / max(ns["vol_attempt"], 1),
                "ms_per_sweep": 1e3 * dt / n_sw,
                "driver": "persistent move server" if st["server_steps"] else "one launch per step",
                "workgroups_per_replica": server_lat_parts(1, nm) // 4,
                "acceptance": (st["trans_accept"] + st["rot_accept"]) / max(st["moves"], 1),
                "volume_acceptance": ns["vol_accept"] / max(ns["vol_attempt"], 1),
                "box_A": ns["box"],
                "energy_drift_rel": abs(e1 - e2) / abs(e2)}
            b.close()
    return out


def call_surface(a, n_moves=2000, n_warm=300):
    """One Markov chain driven through the reference's own calls: the body of Loop()
    (Ewald/main.jl:487-644) -- LJ_poly_dU, EwaldShort, move, LJ_poly_dU, EwaldShort, RecipMove,
    Metropolis, commit or restore -- with the reference's statements and host-array mutations,
    through api.py.  Times every call class with perf_counter around the call."""
    from metropolismontecarlo_amd import api, structs
    from metropolismontecarlo_amd import io as mio
    from metropolismontecarlo_amd.device import Context
    from metropolismontecarlo_amd.structs import EWALD, Properties, Properties2, Tables
    box = a["box"]
    n_mol = a["com"].shape[0]

    def run(server):
        os.environ["MMC_CTX_SERVER"] = "1" if server else "0"
        moa = structs.make_moa(a["com"].copy(), a["first_atom"], a["last_atom"])
        soa = structs.make_soa(a["coords"].copy(), a["atype"], a["charge"])
        vdwTable = Tables([mio.SPCE_EPS_O, 0.0], [mio.SPCE_SIGMA_O, 0.0])
        ewald = EWALD(5.6 / box, 5, 27, 1, [[1, 1, 1]] * 3, [0.0, 0.0], np.zeros(2, complex),
                      np.zeros(2, complex), structs.factor)                   # main.jl:290-301
        ewald = api.PrepareEwaldVariables(ewald, box)                          # main.jl:303
        totProps = Properties2(TEMPERATURE, 0.0331, 0.0, DR_MAX, DPHI_MAX, 0.3, 0, 0, [], RCUT,
                               RCUT, box)
        total = api.potential(moa, soa, Properties(), ewald, vdwTable, totProps, "ewald")
        running = total.energy
        rng = np.random.default_rng(SEED)
        t_call = np.zeros(5)
        t_body = 0.0
        n_acc = 0
        pc = time.perf_counter
        for s_ in range(n_warm + n_moves):
            if s_ == n_warm:
                t_call[:] = 0.0
                t_body = 0.0
                n_acc = 0
                st0 = ewald._session.ctx.stats()
            i = s_ % n_mol + 1                                                 # main.jl:490
            f, l = int(moa.firstAtom[i - 1]), int(moa.lastAtom[i - 1])
            t0 = pc()
            e0, v0 = api.LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)          # :491
            t1 = pc()
            q0, w0, o1 = api.EwaldShort(i, moa, soa, totProps, ewald, box)     # :501
            t2 = pc()
            rm_old = moa.COM[i - 1].copy()                                     # :514
            ra_old = soa.coords[f - 1:l].copy()                                # :515
            d = (rng.random(3) - 0.5) * DR_MAX                                 # :523 (translations)
            moa.COM[i - 1] += d
            soa.coords[f - 1:l] += d                                           # :552
            ra_new = soa.coords[f - 1:l].copy()
            t3 = pc()
            e1, v1 = api.LJ_poly_ΔU(i, moa, soa, vdwTable, RCUT, box)          # :557
            t4 = pc()
            q1, w1, o2 = api.EwaldShort(i, moa, soa, totProps, ewald, box)     # :566
            t5 = pc()
            dr = 0.0
            if not (o1 or o2):
                dr, ewald = api.RecipMove(box, ewald, ra_old, ra_new, soa.charge[f - 1:l])  # :581
            t6 = pc()
            delta = (e1 + q1) - (e0 + q0) + dr                                 # :593
            x = delta / TEMPERATURE
            if (x < 0 or np.exp(-x) > rng.random()) and not (o1 or o2):        # :598
                running += delta
                ewald.sumQExpOld = ewald.sumQExpNew.copy()                     # :621
                n_acc += 1
            else:
                moa.COM[i - 1] = rm_old                                        # :623
                soa.coords[f - 1:l] = ra_old                                   # :624
                ewald.sumQExpNew = ewald.sumQExpOld.copy()                     # :628
            t7 = pc()
            t_call += (t1 - t0, t2 - t1, t4 - t3, t5 - t4, t6 - t5)
            t_body += t7 - t0
        st1 = ewald._session.ctx.stats()
        ping = ewald._session.ctx.ping(2000) if st1["alive"] else None
        total2 = api.potential(moa, soa, Properties(), ewald, vdwTable, totProps, "ewald")
        drift = abs(running - total2.energy) / abs(total2.energy)
        api.release_sessions()
        us = 1e6 * t_call / n_moves
        return {"us_per_call": {"LJ_poly_dU_old": us[0], "EwaldShort_old": us[1],
                                "LJ_poly_dU_new": us[2], "EwaldShort_new": us[3],
                                "RecipMove": us[4]},
                "us_five_calls": float(us.sum()), "us_loop_body": 1e6 * t_body / n_moves,
                "acceptance": n_acc / n_moves, "energy_drift_rel": drift,
                "per_move": {k: (st1[k] - st0[k]) / n_moves
                             for k in ("cmds", "cache_hits", "spec_hits", "spec_miss",
                                       "launch_evals", "look_ahead_hits")},
                "server_round_trip_us": ping}

    out = {"workload": "ONE chain, SPC/E 750 molecules: the body of Loop() (Ewald/main.jl:487-644) "
                       "written with the reference's calls through api.py, one library call each",
           "moves_timed": n_moves}
    out.update(run(True))
    launch = run(False)
    out["launch_per_evaluation"] = {k: launch[k] for k in ("us_per_call", "us_five_calls",
                                                           "us_loop_body", "energy_drift_rel")}
    # the fused form of the same move: mmc_trial_move + mmc_accept_move / mmc_reject_move
    os.environ["MMC_CTX_SERVER"] = "1"
    from metropolismontecarlo_amd import structs as st_
    with Context() as ctx:
        ctx.upload_system(a["com"], a["first_atom"], a["last_atom"], a["coords"], a["atype"],
                          a["charge"], a["eps"], a["sig"], box)
        ctx.prepare_ewald(5.6 / box, 5, 27, box, st_.factor)
        ctx.potential_ewald(RCUT, RCUT)
        com, coords = a["com"].copy(), a["coords"].copy()
        rng = np.random.default_rng(SEED)
        t_tm = 0.0
        for s_ in range(n_warm + n_moves):
            if s_ == n_warm:
                t_tm = 0.0
            i = s_ % n_mol + 1
            d = (rng.random(3) - 0.5) * DR_MAX
            cn, an = com[i - 1] + d, coords[3 * i - 3:3 * i] + d
            t0 = time.perf_counter()
            dd, ov = ctx.trial_move(i, cn, an, RCUT, RCUT)
            x = (dd[0] + dd[1] + dd[2]) / TEMPERATURE
            if (x < 0 or np.exp(-x) > rng.random()) and not ov:
                ctx.accept_move()
                com[i - 1], coords[3 * i - 3:3 * i] = cn, an
            else:
                ctx.reject_move()
            t_tm += time.perf_counter() - t0
        out["us_per_move_trial_move"] = 1e6 * t_tm / n_moves
    os.environ.pop("MMC_CTX_SERVER", None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 600 (3000 below 4096 replicas)")
    ap.add_argument("--warmup", type=int, default=None, help="default 64 (300 below 4096 replicas)")
    ap.add_argument("--replicas", type=int, default=61440,
                    help="replicas per GPU (default: 2 groups x 6 x the 5120 wavefronts the move kernel keeps resident)")
    ap.add_argument("--groups", type=int, default=0,
                    help="replica groups pipelined per GPU (0 = 2, or 1 for a single chain)")
    ap.add_argument("--parts", type=int, default=0, help="units per replica-move (0=auto)")
    ap.add_argument("--threads", type=int, default=0,
                    help="host threads per GPU for the accept/reject (0 = min(8, cores / ranks))")
    ap.add_argument("--kernel", type=int, default=3,
                    help="3 = by launch size (default), 2 = wave per move, 1 = workgroup per move, "
                         "0 = generic")
    ap.add_argument("--wave-wgs", type=int, default=0,
                    help="workgroups of a kernel-2 launch (0 = library default)")
    ap.add_argument("--zero-copy-moves", type=int, default=-1,
                    help="1 = the kernel reads host-written records in place (-1 = only below 4096 replicas)")
    ap.add_argument("--device-moves", type=int, default=1,
                    help="1 = trial moves are drawn on the device (Philox), 0 = by the host driver")
    ap.add_argument("--persistent", type=int, default=-1,
                    help="move server for small batches: -1 = library default (up to one replica per compute unit), "
                         "0 = a launch per step, 1 = insist")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams for the groups (0=auto)")
    ap.add_argument("--accept", choices=("auto", "host", "kernel"), default="auto",
                    help="who takes the Metropolis decision (option accept_on_device): auto = the move "
                         "kernel where a host thread would have more than 4096 records per launch")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the named-config and single-system side measurements")
    ap.add_argument("--no-events", action="store_true",
                    help="do not bracket launches with HIP events in the timed region")
    ap.add_argument("--event-every", type=int, default=8,
                    help="bracket every Nth launch of a group with HIP events (an event pair costs "
                         "~10 us of stream time, so timing every launch slows what it measures)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # one process per GPU.  MMC_DIST_BACKEND=gloo (rehearsal of the N>1 path on a box with fewer
    # GPUs than ranks: ranks then share devices and the reduction runs on CPU tensors)
    backend = os.environ.get("MMC_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    red_device = "cuda" if backend == "nccl" else "cpu"
    # Host threads: every rank spins its own workers (accept/reject, control words through the BAR),
    # so before the first GPU call that creates a thread the rank pins itself to ITS slice of the
    # cores on its GPU's NUMA node (no exec, no re-launch) and sizes its thread count to that slice
    # -- never oversubscribing the node (sharding.plan_host_threads; MMC_NO_PIN=1 keeps the
    # inherited affinity and the old rule min(8, (cores - ranks) / ranks)).
    from metropolismontecarlo_amd import sharding as _sh
    affinity0 = sorted(os.sched_getaffinity(0))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    pinned = None
    if os.environ.get("MMC_NO_PIN"):
        auto_threads = max(1, min(8, (len(affinity0) - local_world) // max(local_world, 1)))
    else:
        try:
            n_dev = torch.cuda.device_count()
            addrs = []
            for k in range(local_world):
                p_ = torch.cuda.get_device_properties(k % n_dev)
                addrs.append((p_.pci_domain_id, p_.pci_bus_id, p_.pci_device_id))
            pinned, auto_threads = _sh.pin_rank_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")) % local_world, addrs)
        except Exception as exc:   # placement is an optimisation: never the reason a run fails
            print(f"bench.py: not pinning host threads ({exc!r})", file=sys.stderr)
            auto_threads = max(1, min(8, (len(affinity0) - local_world) // max(local_world, 1)))
    if args.threads <= 0:
        args.threads = auto_threads
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":  # RCCL over xGMI
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from metropolismontecarlo_amd import io as mio
    from metropolismontecarlo_amd import sharding

    a = mio.load_nist_fixture(4, "unwrapped")
    n_mol, box = a["com"].shape[0], a["box"]
    R = args.replicas

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # chain r of this rank has global index rank*R + r: trajectories depend on the global index
    # only, not on the number of GPUs
    g0 = sharding.shard(R, rank)[0]
    shape = shape_for(R, args)
    res = measure_moves(R, a, args, local_rank, g0, barrier, shape)
    st = res["st"]

    # C1: the only collective -- max of the time, sums of the observables (RCCL all-reduce)
    local = dict(moves=st["moves"], accepted=st["trans_accept"] + st["rot_accept"],
                 overlaps=st["overlaps"], energy_sum=res["energy_sum"],
                 kernel_ms=st["kernel_ms"], launches=st["launches"])
    d = dist if world > 1 else None
    red, elapsed_max = sharding.reduce_observables(local, res["elapsed"], d, device=red_device)
    _, t_full_max = sharding.reduce_observables(local, res["t_full"], d, device=red_device)
    _, drift_max = sharding.reduce_observables(local, res["drift"], d, device=red_device)
    total_moves = red["moves"]

    if rank == 0:
        parts_used = args.parts if args.parts > 0 else default_parts(R, n_mol)
        out = {
            "metric": "MC trial moves/sec (whole node), SPC/E NVT full Ewald fp64",
            "value": total_moves / elapsed_max,
            "unit": "moves/s",
            "n_gpus": world,
            "steps": shape["steps"],
            "warmup": shape["warmup"],
            "ms_per_step": 1e3 * elapsed_max / max(shape["steps"], 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic: NIST SPC/E sample configuration 4 (750 molecules, L=30 A) "
                    "replicated; random trial moves",
            "config": {"workload": "SPC/E 750 molecules NVT 298.15 K, full Ewald (337 k), "
                                   "r_cut 10 A, independent replicas",
                       "replicas_per_gpu": R, "replicas_total": R * world,
                       "groups_per_gpu": shape["groups"], "streams_per_gpu": res["streams"],
                       "host_threads_per_gpu": shape["threads"],
                       "host_cores_pinned": (f"{len(pinned)} cores of the GPU's NUMA node "
                                             f"({pinned[0]}..{pinned[-1]})" if pinned else "inherited affinity"),
                       "prewarm_steps": shape["prewarm"],
                       "move_generation": "device" if args.device_moves else "host",
                       "accept_decision": "move kernel" if st.get("device_decisions", 0) else "host threads",
                       "k_vectors": "337 (ewalds.jl:57-89); the batch's kernels work on 293: one of each "
                                    "exactly-conjugate pair of the kx = 0 plane, twice the weight",
                       **({"short_run_note": f"the timed region of {shape['steps']} steps carries the fill "
                           "and drain of the two-group pipeline (about one kernel time in "
                           f"{2 * shape['steps']}): a 600-step run of the same command reads ~5 % higher"}
                          if shape["steps"] < 100 else {}),
                       "parallelism": f"replicas x{world}"},
            "acceptance": red["accepted"] / max(total_moves, 1),
            "overlaps": int(red["overlaps"]),
            "torn_result_records": int(st["torn_records"]),
            "driver_wall_ms": st["wall_ms"],   # the native driver's own clock over the timed call (this rank)
            "energy_mean_per_replica": red["energy_sum"] / (R * world),
            "energy_drift_rel": drift_max,
            "ns_per_full_energy_eval": 1e9 * t_full_max / R,
            "full_energy_evals_per_s": R * world / t_full_max,
        }
        # M2: a full evaluation is fp64-VALU work, 3.5e7 flop by the survey's
        # counting against 111 KB of compulsory bytes -> fraction of the fp64 vector peak
        flops_full = algorithmic_flops_full_eval(n_mol, box)
        out["full_energy_eval"] = {
            "ns": 1e9 * t_full_max / R, "batched_over_replicas": R,
            "algorithmic_flops": flops_full, "algorithmic_bytes": algorithmic_bytes_full_eval(n_mol),
            "achieved_tflops_per_gpu": flops_full * R / t_full_max / 1e12,
            "frac_fp64_vector_peak_78.6": flops_full * R / t_full_max / 1e12 / FP64_PEAK_TFLOPS}
        spl = max(1, int(round(shape["groups"] * st["moves"] / max(st["launches"], 1) / max(R, 1))))
        out["config"]["driver"] = ("persistent move server (one kernel per run)" if res["server"]
                                   else "one launch per step and group" if spl == 1
                                   else f"one launch per {spl} steps and group: the wave that decides takes a replica "
                                        "through them, one result record per replica and launch")
        rf = (launch_mode_roofline(R, a, args, local_rank, g0, lambda: torch.cuda.synchronize(),
                                   shape, n_mol, box, parts_used) if res["server"]
              else roofline_object(res, R, args, shape, n_mol, box, parts_used))
        if rf:
            out["roofline"] = rf
        # raw work counts: per trial move 2 states x (N_mol - 1) COM tests,
        # 2 x 9 x Mbar atom-pair terms (Mbar = 116.4 neighbours inside the COM gate), 337 x 6 phase
        # terms; launches of the move kernel per move; the driver never synchronises a stream
        mbar = 4.0 / 3.0 * np.pi * RCUT ** 3 * n_mol / box ** 3
        v = out["value"]
        out["work_counts"] = {
            "com_tests_per_s": v * 2 * (n_mol - 1), "atom_pair_terms_per_s": v * 2 * 9 * mbar,
            "lj_pair_terms_per_s": v * 2 * mbar, "phase_terms_per_s": v * N_K * 6,
            "move_kernel_launches_per_move": st["launches"] / max(st["moves"], 1),
            "stream_syncs_per_move": 0.0,
            "pcie_bytes_per_move": {"h2d": (0 if st.get("device_decisions") else 1) if args.device_moves else 232,
                                    "d2h": 64 * parts_used / spl}}
        if not args.no_secondary and world == 1:   # side measurements: single-GPU runs only
            # BASELINE's two named replica counts on this GPU, each with its own roofline object:
            # configs[1] = one chain (latency-bound), configs[2] = 256 replicas over 8 GPUs = 32/GPU
            out["named_configs"] = {}
            for name, r2 in (("configs[1]: 1 replica on 1 GPU", 1),
                             ("configs[2] share: 32 replicas per GPU", 32)):
                if r2 == R:
                    continue
                sh2 = shape_for(r2, argparse.Namespace(**{**vars(args), "steps": None, "warmup": None,
                                                          "groups": 0, "zero_copy_moves": -1}))
                r_ = measure_moves(r2, a, args, local_rank, 0, barrier, sh2, n_parts=0)
                s2 = r_["st"]
                entry = {"moves_per_s_per_gpu": s2["moves"] / r_["elapsed"],
                         "us_per_step": 1e6 * r_["elapsed"] / sh2["steps"],
                         "us_per_move_per_chain": 1e6 * r_["elapsed"] / sh2["steps"],
                         "steps": sh2["steps"], "groups": sh2["groups"],
                         "energy_drift_rel": r_["drift"],
                         "ns_per_full_energy_eval": 1e9 * r_["t_full"] / r2}
                entry["driver"] = ("persistent move server" if r_["server"]
                                   else "one launch per step and group")
                rf2 = (launch_mode_roofline(r2, a, args, local_rank, 0, barrier, sh2, n_mol, box,
                                            default_parts(r2, n_mol)) if r_["server"]
                       else roofline_object(r_, r2, args, sh2, n_mol, box, default_parts(r2, n_mol)))
                entry["units_per_move"] = ((server_lat_parts(r2, n_mol) or (5 if r2 == 1 else 8))
                                           if r_["server"] else default_parts(r2, n_mol))
                if rf2:
                    entry["roofline"] = rf2
                out["named_configs"][name] = entry
            if R >= 4096 and args.streams == 0 and not res["server"] and res["streams"] > 1:
                # The same workload with every launch ALONE on the GPU (one stream): event time,
                # rocprofv3 trace time and step time then add up, and the kernel's own duration can
                # be read -- but the start of every launch (5120 waves in step) and its end (a
                # thinning tail) are paid in full: the headline's two streams overlap them.
                # Same pre-warm as the headline (a cold side run was what made round 3's
                # `two_streams` line read 6.7e7 in the driver's run).
                a1 = argparse.Namespace(**{**vars(args), "streams": 1})
                sh1 = dict(shape, steps=min(max(shape["steps"], 100), 200), warmup=min(shape["warmup"], 24),
                           prewarm=104)
                r_ = measure_moves(R, a, a1, local_rank, g0, barrier, sh1)
                rf1 = roofline_object(r_, R, a1, sh1, n_mol, box, parts_used)
                out["one_stream"] = {
                    "value": r_["st"]["moves"] / r_["elapsed"], "unit": "moves/s",
                    "ms_per_step": 1e3 * r_["elapsed"] / sh1["steps"], "steps": sh1["steps"],
                    "avg_launch_us": rf1["avg_launch_us"] if rf1 else None,
                    "avg_launch_us_is": rf1["avg_launch_us_is"] if rf1 else None,
                    "frac": rf1["frac"] if rf1 else None,
                    "binding_frac": rf1.get("binding", {}).get("frac") if rf1 else None,
                    "binding_bound": rf1.get("binding", {}).get("bound") if rf1 else None,
                    "energy_drift_rel": r_["drift"],
                    "note": "every launch alone on the GPU: its HIP-event duration is its cost and is what "
                            "rocprofv3 --kernel-trace reports for `--streams 1` (profiles/)"}
            if args.accept != "host" and st.get("device_decisions"):
                # The same chains with the accept decision on the HOST (north_star's placement: its
                # threads read a record per replica and step, decide, and send a flag byte back),
                # one step per launch: same seeds, the same decisions bit for bit.
                ah = argparse.Namespace(**{**vars(args), "accept": "host"})
                shh = dict(shape, steps=min(max(shape["steps"], 100), 200), warmup=min(shape["warmup"], 24),
                           prewarm=104)
                rh = measure_moves(R, a, ah, local_rank, g0, barrier, shh)
                rfh = roofline_object(rh, R, ah, shh, n_mol, box, parts_used)
                out["host_decides"] = {
                    "value": rh["st"]["moves"] / rh["elapsed"], "unit": "moves/s",
                    "ms_per_step": 1e3 * rh["elapsed"] / shh["steps"], "steps": shh["steps"],
                    "avg_launch_us": rfh["avg_launch_us"] if rfh else None,
                    "moves_per_launch": rfh["moves_per_launch"] if rfh else None,
                    "acceptance": (rh["st"]["trans_accept"] + rh["st"]["rot_accept"]) / max(rh["st"]["moves"], 1),
                    "energy_drift_rel": rh["drift"],
                    "note": "`--accept host`: the sequential accept/reject on the host's threads, one launch "
                            "and one 64-byte record per replica per step; the headline lets the move kernel "
                            "take the same decision (same Philox uniform, same arithmetic) and eight steps "
                            "per launch"}
            out["full_energy_eval"]["single_system_latency"] = single_system_latency(a, local_rank)
            out["call_surface"] = call_surface(a)
        if not args.no_cpu and world == 1:         # the CPU baseline leg: rank 0 at N=1 only
            os.sched_setaffinity(0, affinity0)     # all cores of the host again
            mps, n, dt, tf = cpu_baseline(a, args.cpu_seconds)
            out["cpu_baseline"] = {
                "value": mps, "unit": "moves/s", "cores": 1, "kind": "port",
                "sample": f"{n} trial moves (2x LJ_poly_dU + 2x EwaldShort + RecipMove) of the "
                          f"same 750-molecule system in {dt:.1f} s, C oracle (C loop), 1 thread",
                "ns_per_full_energy_eval": 1e9 * tf,
                "us_per_move": 1e6 / mps,
            }
            if "call_surface" in out:
                out["call_surface"]["cpu_port_us_per_move"] = 1e6 / mps
            nt = min(len(os.sched_getaffinity(0)), 64)
            if nt > 1 and args.cpu_seconds >= 2:
                mps_all, n_all, dt_all, _ = cpu_baseline(a, args.cpu_seconds / 2, nt)
                out["cpu_baseline_all_cores"] = {
                    "value": mps_all, "unit": "moves/s", "cores": nt, "kind": "port",
                    "sample": f"{n_all} trial moves, one independent chain per thread, "
                              f"{dt_all:.1f} s"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
