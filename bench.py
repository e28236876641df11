#!/usr/bin/env python3
"""bench.py — Mray/s (primary + secondary rays) of the wavefront path tracer on the BASELINE.json configurations.

  N = 1  the headline configuration (configs[2]): S3 "ganesha-proxy" (4 305 626 primitives, 8.52 M BVH nodes), 1024x1024,
         256 spp, maxdepth 5. A step = one whole ImageTileIntegrator::render of that frame (all spp-waves over all 8x8 tiles).
  N > 1  the scaling configuration (configs[4], "C5"): the same scene at 3840x2160, 1024 spp, tiles sharded across the N GPUs
         inside the library (shm_render_sharded: C++ tile sharding, no collective while rendering, RCCL gather of the film rows
         to rank 0 inside the timed region). One process per GPU, started EITHER by `python -m torch.distributed.run ... bench.py
         --gpus N` (the driver's way: RANK / LOCAL_RANK / WORLD_SIZE come from the environment) OR by a plain `python bench.py --gpus N`:
         the parent then starts N fresh children itself before making any GPU call (shimmer_amd/launch.py), relays rank 0's JSON line
         and exits non-zero if any child does.
A rank process imports NO torch: it maps one HIP runtime and one RCCL, the ones libshimmer_hip.so is linked against (/opt/rocm). The
128-byte RCCL id travels from rank 0 to the others through a directory of files; the barrier either side of the timed region, the
max-over-ranks clock and the counters run through the library's own collectives (shm_dist_barrier / _allreduce_f64 / _allgather_f64).
Scene arrays are resident in HBM before the timed region. --width/--height/--spp override either default.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--spp S] [--res R] [--no-cpu-baseline] [--no-side] [--no-live-pmc]
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # MI355X_MICROARCH.md: measured-achievable copy rate
# FETCH_SIZE -> bytes: the guide's gfx950 correction (x2) is calibrated there for wide coalesced streams; tools/ubench/gather_fetch.hip measures it for
# this kernel's pattern (random pairs of 16-B loads from 32-B records of an array far larger than the 256 MiB Infinity Cache): profiles/r04_gather_fetch.txt
FETCH_FACTOR = 2.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"),
              ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY"))
# measured on this part (tools/ubench/valu_rate.hip -> profiles/r03_valu_issue_rates.txt, >= 2 waves per SIMD): clocks a SIMD is held per wave64 VALU
# instruction. The cheapest class prices the CEILING (no instruction mix can issue faster); the counter's own pricing is one quad-cycle = 4 clocks.
VALU_CLOCKS_BEST = 2.4     # v_add / v_mul / v_fma / v_add_u32 / v_mov
VALU_CLOCKS_GUIDE = 2.0    # MI355X_MICROARCH.md "Per-instruction cycle constants": v_fma_f32 (wave64) = 2 cyc (SIMD-32) — the guide's ceiling, stated beside the measured one
VALU_CLOCKS_COUNTER = 4.0  # what SQ_ACTIVE_INST_VALU charges; v_cndmask (SGPR mask) / v_cmp -> SGPR / v_max3 measure 4.2
N_SIMD, N_XCD = 256 * 4, 8
K2_NAME = "k_trace<closest>"  # k_trace5<false, GEN> (the both-children step; GEN: scenes with spheres / patches / instances)


def _counters_from_db(db, want):
    """Per-dispatch means of every counter of one kernel in a rocprofv3 rocpd database, and its mean duration there. `want`: "closest" — the
    closest-hit traversal kernel (k_trace5<false, *>) — or "shade" — the fused shading kernel of the headline scene class."""
    import re
    import sqlite3
    cur = sqlite3.connect(db).cursor()
    acc, disp = {}, {}
    for name, counter, value, d, dur in cur.execute("select kernel_name, counter_name, value, dispatch_id, duration from counters_collection"):
        if want == "closest":
            m = re.search(r"k_trace\d<(\w+)", name)  # (k_trace5<ANY, GEN, HEAVY>: the first argument decides)
            if not m or m.group(1) != "false":  # first template argument: ANY
                continue
        elif not re.search(r"k_shade<false, true, false, true\b", name):
            continue
        acc[counter] = acc.get(counter, 0.0) + float(value)
        disp[d] = float(dur)
    n = max(1, len(disp))
    out = {k: v / n for k, v in acc.items()}
    if disp:
        out["_duration_ms"] = sum(disp.values()) / len(disp) / 1e6
    return out, len(disp)


def live_pmc(args):
    """Collects the dominant kernel's hardware counters IN THIS RUN: one child process per counter set — `rocprofv3 --pmc <set> -- python3
    bench.py --pmc-child ...` renders one frame of the same configuration — started before this process makes any GPU call. Separate
    passes, no trace domain beside --pmc (MI355X_MICROARCH.md "rocprofv3 PMC slots"). Returns {counter: mean per K2 launch} or None."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out, t0 = {}, time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="shm_pmc_")
    try:
        child = [sys.executable, str(Path(__file__).resolve()), "--pmc-child", "--spp", str(args.spp), "--res", str(args.res), "--n", str(args.n),
                 "--max-depth", str(args.max_depth), "--seed", str(args.seed), "--steps", "1", "--warmup", "0"]
        for i, counters in enumerate(PMC_PASSES):
            d = os.path.join(tmp, f"pass{i}")
            cmd = [exe, "--pmc", *counters, "-d", d, "-o", "pmc", "--"] + child
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            r = subprocess.run(cmd, env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.pmc_timeout)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, f"pass {counters[0]}: rc {r.returncode}: {r.stderr.decode(errors='replace')[-300:]}"
            vals, n = _counters_from_db(dbs[0], "closest")
            if not n:
                return None, f"pass {counters[0]}: no {K2_NAME} dispatch in the database"
            out.update(vals)
            out["dispatches"] = n
            sv, sn = _counters_from_db(dbs[0], "shade")  # (the fused vertex kernel, second to the closest-hit traversal in the frame: the same accounting, informational)
            if sn:
                out.setdefault("_shade", {}).update(sv)
                out["_shade"]["dispatches"] = sn
        out["collect_s"] = time.perf_counter() - t0
        return out, "live: rocprofv3 --pmc passes of this run"
    except Exception as e:  # reporting only
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def committed_pmc(args):
    """Fallback: the committed rocprofv3 PMC passes of THIS configuration (tools/profile_gpu.sh -> profiles/traffic_*.json)."""
    f = ROOT / "profiles" / f"traffic_res{args.res}_spp{args.spp}_n{args.n}_depth{args.max_depth}.json"
    if not f.exists() or args.width or args.height:
        return None
    j = json.loads(f.read_text())
    t = j.get("k_trace5<closest>") or j.get("k_trace3<closest>")
    if not t:
        return None
    c = {"FETCH_SIZE": t.get("FETCH_SIZE_bytes_per_dispatch_raw", 0.0) / 1024.0, "WRITE_SIZE": t.get("WRITE_SIZE_bytes_per_dispatch_raw", 0.0) / 1024.0}
    if t.get("valu_lanes_active"):
        c["lanes"] = t["valu_lanes_active"]
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SALU"):
        if k in t:
            c[k] = t[k]
    return c


def counter_blocks(c, avg_launch_ms):
    """From the dominant kernel's counters (means per launch): HBM traffic (2 x FETCH_SIZE — the gfx950 correction of MI355X_MICROARCH.md
    section HBM, calibrated for THIS access pattern by tools/ubench/gather_fetch.hip — + WRITE_SIZE, KiB units), lanes per VALU instruction,
    and the VALU-issue roofline: lane-operations per second against what 1024 SIMDs x 64 lanes can issue."""
    if not c:
        return None, None, None
    traffic = (FETCH_FACTOR * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0 if "FETCH_SIZE" in c else None
    lanes = c.get("lanes")
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        lanes = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
    valu = None
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE") and lanes and avg_launch_ms:
        cycles = c["GRBM_GUI_ACTIVE"] / N_XCD                       # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        clock_hz = cycles / (avg_launch_ms * 1e-3)
        n_inst = c.get("SQ_INSTS_VALU") or c["SQ_ACTIVE_INST_VALU"]  # (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.00 in this kernel: one quad-cycle each)
        lane_ops = n_inst * lanes                                   # = SQ_THREAD_CYCLES_VALU when every instruction is charged one quad-cycle
        achieved = lane_ops / (avg_launch_ms * 1e-3)                # lane-operations per second
        peak = N_SIMD * 64 * clock_hz / VALU_CLOCKS_BEST            # every SIMD issuing a full-width 2.4-clock instruction back to back
        busy4 = VALU_CLOCKS_COUNTER * c["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cycles)
        valu = {"kernel": "k_trace<closest>", "achieved_lane_ops_per_s": achieved, "peak_lane_ops_per_s": peak, "frac": achieved / peak,
                "frac_at_counter_pricing_4_clocks": achieved / (N_SIMD * 64 * clock_hz / VALU_CLOCKS_COUNTER),
                "frac_at_guide_peak_2_clocks": achieved / (N_SIMD * 64 * clock_hz / VALU_CLOCKS_GUIDE),
                "peak_lane_ops_per_s_guide_2_clocks": N_SIMD * 64 * clock_hz / VALU_CLOCKS_GUIDE,
                "valu_busy_frac_at_4_clocks": busy4, "valu_busy_frac_at_2p4_clocks": busy4 * VALU_CLOCKS_BEST / VALU_CLOCKS_COUNTER,
                "lanes_per_valu_inst": lanes, "valu_insts_per_launch": n_inst, "gpu_cycles_per_launch": cycles, "effective_clock_GHz": clock_hz / 1e9,
                "issue_rate_measured_clocks_per_inst": {"simple f32 / int (add, mul, fma, mov)": 2.4, "select / compare-to-SGPR / packed / 3-operand": 4.2, "rcp": 8.2},
                "wait_frac": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES") else None,
                "wait_inst_frac": (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAIT_INST_ANY") and c.get("SQ_WAVE_CYCLES") else None,
                "salu_busy_frac_at_4_clocks": (4.0 * c["SQ_ACTIVE_INST_SCA"] / (N_SIMD * cycles)) if c.get("SQ_ACTIVE_INST_SCA") else None,
                "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
                "definitions": "achieved = SQ_INSTS_VALU x (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU) / launch time = active-lane VALU operations per second; "
                               "peak = 1024 SIMDs x 64 lanes x measured clock (GRBM_GUI_ACTIVE / 8 XCDs / launch time) / 2.4 clocks — the fastest rate a SIMD issues "
                               "wave64 VALU instructions at (profiles/r03_valu_issue_rates.txt), so no instruction mix can exceed it; the slab test is mostly 4.2-clock "
                               "selects and compares, for which frac_at_counter_pricing_4_clocks is the closer figure. frac = lane occupancy x issue-slot occupancy: "
                               "lanes_per_valu_inst / 64 x valu_busy_frac. wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (the share of a wave's resident cycles it spends waiting — mostly on "
                               "its dependent gathers), wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (waiting for an instruction to issue); SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, 's_memtime tick vs SQ PMC units')"}
    return traffic, lanes, valu


def cpu_baseline(sc, params_full, host_lib, budget_s=15.0):
    """The CPU oracle (kind "port": a C++ restatement of the reference loop, oracle/oracle.cpp) timed on this box's host cores on a
    bounded UNBIASED sample of the same workload: every k-th 8x8 tile of the whole frame (row-major tile order, so object, walls
    and floor are sampled in the frame's own proportions) at the full spp."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle_py
    from shimmer_amd import abi, render, scene as scn

    cores = os.cpu_count() or 1
    orc = oracle_py.Oracle(sc.desc)
    tiles, n_tiles = scn.tiles_for(host_lib, orc.pixel_bounds)

    def run(stride, spp, offset=0):
        idx = list(range(offset, n_tiles, stride))
        sub = (abi.ShmTile * len(idx))(*[tiles[i] for i in idx])
        p = render.make_params(seed=params_full.seed, spp=spp, max_depth=params_full.max_depth)
        t0 = time.perf_counter()
        _, st = orc.render(p, n_threads=cores, tiles=sub, n_tiles=len(idx))
        return len(idx), st["rays_closest"] + st["rays_any"], time.perf_counter() - t0

    # probe: 16 and 64 spp on every 61st tile (61 is coprime with the tiles per row: no column aliasing) sizes the sample to ~budget_s. (Round 4: the probe
    # used to be 1 and 5 spp — 0.03 s on 256 threads, all start-up cost: the difference was noise and the "sample" became the whole frame, 97 s.)
    run(61, 1)  # thread start-up, page faults
    n_probe, _, t0 = run(61, 16)
    _, rays1, t1 = run(61, 64)
    full = params_full.samples_per_pixel
    # seconds per tile per spp: the difference (the fixed per-call cost cancels), but never less than 0.7 x the plain rate of the larger probe
    per_tile_spp = max((t1 - t0) / 48.0, 0.7 * t1 / 64.0, 1e-5) / n_probe
    want_tiles = budget_s / (per_tile_spp * full)
    stride = int(max(1, min(n_tiles, round(n_tiles / max(want_tiles, 1.0)))))
    while stride > 1 and stride % 2 == 0 and ((orc.width + 7) // 8) % 2 == 0:
        stride += 1  # keep the stride odd when the tile row length is even: the sample walks across columns
    n_used, rays, dt = run(stride, full)
    orc.close()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": f"oracle/oracle.cpp, {cores} threads over 8x8 tiles; every {stride}th tile of the whole frame ({n_used} of {n_tiles} tiles, "
                      f"unbiased tile sample) at the full {full} spp ({rays} rays in {dt:.1f} s; probe {rays1} rays in {t1:.2f} s); cpu: {model}"}


def side_results(lib, args, render, scenes, headline_scene, log):
    """Other BASELINE.json configurations and the material the real Ganesha uses, timed on this GPU in the same run (one warm-up +
    two timed frames each, their mean reported; informational, outside `value`)."""
    import ctypes as C
    out = {}

    def timed(name, desc, spp, depth, prims, **params):
        r = render.Renderer(lib, desc, device=0)
        p = render.make_params(seed=0, spp=spp, max_depth=depth, **params)
        r.clear()
        r.render_device(p)
        frames = []
        for _ in range(2):  # one warm-up + TWO timed frames (round 6: single frames differ by +-2 % run to run; both are in the line, the mean is the figure)
            r.clear()
            t0 = time.perf_counter()
            st = r.render_device(p)
            frames.append(time.perf_counter() - t0)
        dt = sum(frames) / len(frames)
        r.close()
        rays = st["rays_closest"] + st["rays_any"]
        out[name] = {"Mray_s": rays / dt / 1e6, "ms": dt * 1e3, "ms_frames": [f * 1e3 for f in frames], "rays": rays, "prims": prims, "spp": spp, "max_depth": depth,
                     "ms_closest": st["ms_trace_closest"], "ms_any": st["ms_trace_any"], "ms_shade": st["ms_shade"]}
        log(f"[bench] side {name}: {rays / dt / 1e6:.0f} Mray/s ({dt * 1e3:.1f} ms)")

    try:
        # S3 with the object's material switched to CoatedDiffuse in place (the reference's Ganesha render uses it): same geometry, BVH
        # the headline frame with ShmRenderParams::disable_reference_quirks (PBRT-v4's forms of the reference's deviations): the same kernels, the
        # same speed — the switch costs a host nothing
        timed("S3_reference_quirks_off_1024x1024_spp256", headline_scene.desc, 256, args.max_depth, headline_scene.info["n_primitives"], reference_quirks=False)
        b = headline_scene.builder
        tmp = type(b)()
        tmp.material_coated_diffuse(reflectance=0.4, roughness=0.05, thickness=0.01)
        saved = type(tmp.materials[0])()
        C.memmove(C.byref(saved), C.byref(headline_scene.desc.materials[0]), C.sizeof(saved))
        C.memmove(C.byref(headline_scene.desc.materials[0]), C.byref(tmp.materials[0]), C.sizeof(saved))
        try:
            timed("coated_S3_1024x1024_spp64", headline_scene.desc, 64, args.max_depth, headline_scene.info["n_primitives"])
            # ... and at the headline's own sample count: the headline frame with the real scene's material
            timed("coated_S3_1024x1024_spp256", headline_scene.desc, 256, args.max_depth, headline_scene.info["n_primitives"])
        finally:
            C.memmove(C.byref(headline_scene.desc.materials[0]), C.byref(saved), C.sizeof(saved))
        # the headline frame with the shapes a real PBRT-v4 scene mixes into its triangles (round 5): the window emitter as ONE bilinear patch (what a quad PLY
        # face becomes, shape/shape.rs:119-134), a sphere beside the object, the object as a TransformedPrimitive (primitive.rs:136-176)
        # ... and the object under an ImageInfinitelight (light.rs:805-981): escaped rays look the map up, next-event estimation samples its distribution
        # (round 6: "quads" — the object's cells as 2.15 M bilinear patches instead of 4.3 M triangles: what a quad PLY file, the reference's showcase Ganesha, becomes there
        #  (shape/mesh.rs:233-256): every leaf of the object a non-triangle test — the traversal kernels' five-wave instantiations)
        for variant, name in (("patch_emitter", "S3_patch_emitter"), ("one_sphere", "S3_with_one_sphere"), ("instanced", "S3_instanced"), ("environment", "S3_environment_map"),
                              ("textured_floor", "S3_textured_floor"),  # (ONE textured material among plain ones: the split pass in front of the textured kernels)
                              ("quads", "S3_as_bilinear_patches"),
                              ("instance_grid", "S3_as_64_instances")):  # (one small definition placed 4 x 4 x 4 times: rays cross several instance boxes, more node visits per ray)
            sc = scenes.ganesha_proxy(lib, 1024, 1024, n=args.n, variant=variant)
            timed(f"{name}_1024x1024_spp256", sc.desc, 256, args.max_depth, sc.info["n_primitives"])
            del sc
        # ... the showcase's own class: a CoatedDiffuse object made of bilinear patches
        sc = scenes.ganesha_proxy(lib, 1024, 1024, n=args.n, coated=True, variant="quads")
        timed("coated_S3_as_bilinear_patches_1024x1024_spp256", sc.desc, 256, args.max_depth, sc.info["n_primitives"])
        del sc
        # ... and the reference's showcase class: the coated object under the map (the staged kernels' K_ENV_LIGHT units)
        sc = scenes.ganesha_proxy(lib, 1024, 1024, n=args.n, coated=True, variant="environment")
        timed("coated_S3_environment_map_1024x1024_spp256", sc.desc, 256, args.max_depth, sc.info["n_primitives"])
        del sc
        sc = scenes.crown_proxy(lib, 1000, 1400)
        timed("C4_crown_proxy_1000x1400_spp256_depth32", sc.desc, 256, 32, sc.info["n_primitives"])
        c4 = out["C4_crown_proxy_1000x1400_spp256_depth32"]
        try:  # "mean path length; queue occupancy" — one more, UNTIMED render of the same frame in a child process with SHM_DEBUG=1: the library reads its queue counters back after every bounce
            import subprocess
            r = subprocess.run([sys.executable, str(ROOT / "tools" / "c4_bounces.py"), "--json"], capture_output=True, text=True, timeout=300)
            c4.update(json.loads(r.stdout.strip().splitlines()[-1]))
        except Exception as e:  # reporting only
            c4["per_bounce_error"] = f"{type(e).__name__}: {e}"
        # glass and metal under an ImageInfinitelight beside the emitter (round 5: the sorted fused kernel's ENV_LIGHT instantiation instead of the textured class)
        sc = scenes.crown_proxy(lib, 1000, 1400, environment=scenes.environment_image(64))
        timed("C4_crown_proxy_environment_map_1000x1400_spp256_depth32", sc.desc, 256, 32, sc.info["n_primitives"])
        sc = scenes.cornell_box(lib, 512, 512)
        timed("C2_cornell_512x512_spp64", sc.desc, 64, 5, sc.info["n_primitives"])
        sc = scenes.cornell_box(lib, 512, 512, textured=True)
        timed("textured_cornell_512x512_spp64_depth6", sc.desc, 64, 6, sc.info["n_primitives"])
    except Exception as e:  # reporting only
        out["error"] = str(e)
    return out


_STORE = None  # this rank's control-channel store (so that main() can publish a failure)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0, help="default: 256 at N = 1 (headline), 1024 at N > 1 (C5)")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--width", type=int, default=0, help="non-square frames: overrides --res (default at N > 1: 3840x2160, BASELINE config C5)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--n", type=int, default=599, help="cube-sphere subdivision (599 -> 4 305 612 triangles)")
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the side configurations (coated S3, C4, C2, textured Cornell)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not collect the dominant kernel's counters in this run (falls back to the committed profile)")
    ap.add_argument("--pmc-timeout", type=float, default=240.0, help="seconds per live counter pass")
    ap.add_argument("--pmc-child", action="store_true", help="internal: the frame a live counter pass profiles (no baseline, no side runs, prints nothing)")
    ap.add_argument("--coated", action="store_true", help="S3 with a CoatedDiffuse object (LayeredBxDF, SURVEY 8f-1) instead of the "
                    "headline diffuse one: a side measurement, not the BASELINE config")
    ap.add_argument("--variant", default=None, choices=["patch_emitter", "one_sphere", "instanced", "environment", "textured_floor", "quads"],
                    help="development: S3 with a bilinear-patch emitter / one sphere / the object instanced (the side results' scenes) instead of the headline scene")
    ap.add_argument("--shard-of", type=int, default=0, help="development: render only the tiles rank 0 of N would own (no gather), "
                    "to estimate the per-rank time of an N-GPU run on one GPU")
    ap.add_argument("--shard-rank", type=int, default=0, help="development: which rank's tiles --shard-of renders")
    ap.add_argument("--force-dist", action="store_true", help="exercise the RCCL communicator + shm_render_sharded path even with one rank")
    ap.add_argument("--launch", action="store_true", help="go through the self-launcher even for N = 1 (one fresh child process)")
    ap.add_argument("--launch-timeout", type=float, default=0.0, help="seconds the launcher waits for its ranks (0 = no limit)")
    ap.add_argument("--init-timeout", type=float, default=600.0, help="seconds a rank waits for the id exchange + ncclCommInitRank before it gives up")
    ap.add_argument("--dry-run", action="store_true", help="control plane only (no GPU, no library render call): launcher, environment, id exchange, "
                    "barrier and max-reduce through the store; used by the CPU tests")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="with --dry-run: this rank exits 7 after the id exchange")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    in_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not in_rank and (args.gpus > 1 or args.launch):
        # the parent of a plain `python bench.py --gpus N`: N fresh children before any GPU call; it relays rank 0's line
        from shimmer_amd import launch
        log(f"[bench] self-launch: {args.gpus} rank process(es), one per GPU (no torch.distributed.run above us)")
        argv = [sys.executable, str(Path(__file__).resolve())] + [a for a in sys.argv[1:] if a != "--launch"]
        sys.exit(launch.spawn_ranks(argv, args.gpus, timeout_s=args.launch_timeout or None))
    try:
        rank_main(args)
    except BaseException as e:  # a failing rank tells its peers through the store before it dies, so that nobody waits for it
        clean_exit = isinstance(e, SystemExit) and e.code in (0, None)
        if _STORE is not None and not clean_exit:
            _STORE.fail(f"{type(e).__name__}: {e}")
        raise


def rank_main(args):
    global _STORE
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    c5 = world > 1 and not (args.width or args.height)
    if c5:
        args.width, args.height = 3840, 2160
    if not args.spp:
        args.spp = 1024 if world > 1 else 256
    use_dist = world > 1 or args.force_dist
    from shimmer_amd import launch
    store = None
    if use_dist or args.dry_run:
        store = _STORE = launch.store_from_env(rank, world, timeout_s=args.init_timeout)

    if args.dry_run:
        return dry_run(args, rank, local_rank, world, store)

    headline = world == 1 and not (args.coated or args.variant or args.shard_of or args.force_dist or args.width or args.height or args.pmc_child)
    counters, counters_src = None, "none"
    if headline and not args.no_live_pmc:
        # BEFORE this process touches the GPU: the counter passes are child processes under rocprofv3
        counters, counters_src = live_pmc(args)
        log(f"[bench] live counters: {counters_src}" + (f" ({counters['collect_s']:.0f} s)" if counters else ""))
    if counters is None and world == 1:
        c = committed_pmc(args)
        if c:
            counters, counters_src = c, f"committed profile profiles/traffic_res{args.res}_spp{args.spp}_n{args.n}_depth{args.max_depth}.json" + \
                (f" (live collection failed: {counters_src})" if counters_src not in ("none",) else "")

    import numpy as np
    from shimmer_amd import abi, scenes, render

    lib = abi.load_library()
    if "torch" in sys.modules:
        raise SystemExit("bench.py must not import torch: the rank process maps ONE HIP runtime and ONE RCCL (the library's)")
    if lib.shm_device_count() <= local_rank:
        raise SystemExit(f"bench.py needs HIP device {local_rank}: the render path has no CPU fallback ({lib.shm_device_count()} visible)")

    def device_sync():
        abi.check(lib, lib.shm_device_synchronize(local_rank), "shm_device_synchronize")

    t0 = time.perf_counter()
    width, height = (args.width or args.res), (args.height or args.res)
    sc = scenes.ganesha_proxy(lib, width, height, n=args.n, coated=args.coated, variant=args.variant)
    t_scene = time.perf_counter() - t0
    t0 = time.perf_counter()
    r = render.Renderer(lib, sc.desc, device=local_rank)
    t_upload = time.perf_counter() - t0
    if rank == 0:
        log(f"[bench] scene {sc.name}: {sc.info['n_primitives']} prims, {sc.info['n_nodes']} nodes; build {t_scene:.1f}s, upload {t_upload:.2f}s")
    params = render.make_params(seed=args.seed, spp=args.spp, max_depth=args.max_depth)
    if use_dist:
        # the library's own RCCL communicator: rank 0 draws the unique id, the file store carries its 128 bytes; a rank that cannot get
        # this far publishes its failure in the store (main()), and a watchdog bounds ncclCommInitRank, which waits for every rank
        # (RCCL prints its version banner through C stdio on stdout when the communicator is made: stdout belongs to the ONE JSON line, so
        #  file descriptor 1 points at stderr while the communicator is being set up)
        import ctypes
        libc = ctypes.CDLL(None)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            with launch.Watchdog(args.init_timeout, "id exchange + shm_dist_init (ncclCommInitRank)", store):
                uid = store.broadcast("rccl_unique_id", r.dist_unique_id() if rank == 0 else None)
                r.dist_init(rank, world, uid)
                r.dist_barrier()  # the first collective, still with the banner going to stderr
        finally:
            libc.fflush(None)
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    info = r.dist_info()
    if rank == 0:
        log(f"[bench] runtime: librccl {info['librccl_path']} (RCCL {info['rccl_version']}), libamdhip64 {info['libamdhip_path']} (HIP {info['hip_runtime_version']}); "
            f"communicator: {info['rccl_ranks']} rank(s), world {world}, {info['rows_per_block']} tile rows per shard block")
    my_tiles = None
    if world == 1 and args.shard_of > 1:
        my_tiles = render.shard_tiles(r.n_tiles, r.tiles_per_row, args.shard_rank, args.shard_of, lib=lib)

    def barrier():
        r.dist_barrier()   # drains this rank's streams, then (with a communicator) one all-reduce over RCCL: every rank arrived
        device_sync()

    def step():
        if use_dist:
            return r.render_sharded(params)  # clear + this rank's tiles + RCCL gather of the film rows into rank 0's device film
        r.clear()
        return r.render_device(params, my_tiles)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        st = step()
        for k, v in st.items():
            acc[k] = acc.get(k, 0) + v
    barrier()
    dt = time.perf_counter() - t0
    if args.pmc_child:
        r.close()
        return
    per_rank = None
    if use_dist:
        dt = r.dist_allreduce([dt], abi.SHM_REDUCE_MAX)[0]   # the slowest rank's clock
        keys = ["paths", "rays_closest", "rays_any", "nodes_closest", "tris_closest", "nodes_any", "tris_any"]
        tot = dict(zip(keys, r.dist_allreduce([float(acc[k]) for k in keys], abi.SHM_REDUCE_SUM)))
        mine = [acc["ms_total"] / args.steps, acc["ms_gather"] / args.steps, float(acc["rays_closest"] + acc["rays_any"]) / args.steps,
                acc["gather_bytes"] / args.steps / 1e6, float(info["n_my_tiles"]), float(info["rccl_device"])]
        allr = r.dist_allgather(mine)
        per_rank = [{"rank": i, "render_ms": v[0], "gather_ms": v[1], "rays_per_step": v[2], "gather_MB": v[3], "tiles": int(v[4]), "device": int(v[5])}
                    for i, v in enumerate(allr)]
    else:
        tot = {k: float(v) for k, v in acc.items()}

    out, frac_fatal = None, False
    if rank == 0:
        rays = tot["rays_closest"] + tot["rays_any"]
        value = rays / dt / 1e6
        # roofline of the dominant kernel (K2 trace_closest) on rank 0, from HIP events on the library's render stream:
        # algorithmic bytes = 32 B per node visited + 48 B per primitive tested + (32 B ray read + 16 B hit write) per ray
        # (SURVEY §8d), summed over this rank's launches, over the summed launch durations (= per-launch averages' ratio).
        # (48 B per primitive is the survey's convention — three vertices padded —, not the size of the device record: PrimRec is one aligned 64-byte line, shm/scene.h)
        bytes_alg = 32.0 * acc["nodes_closest"] + 48.0 * acc["tris_closest"] + 48.0 * acc["rays_closest"]
        ms = acc["ms_trace_closest"]
        achieved = bytes_alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        launches = max(1, acc["launches_closest"])
        traffic, lanes, valu = counter_blocks(counters, ms / launches)
        hbm_counter = traffic / (ms / launches * 1e-3) / 1e9 if traffic and ms > 0 else None
        out = {
            "metric": "Mray/s (primary+secondary) at 1024^2 256spp Ganesha; 1/2/4/8-GPU scaling",
            "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'S3c coated ' if args.coated else 'S3 '}{'[' + args.variant + '] ' if args.variant else ''}ganesha-proxy ({sc.info['n_primitives']} prims, {sc.info['n_nodes']} BVH nodes), "
                                   f"{width}x{height}, {args.spp} spp, maxdepth {args.max_depth}, path integrator"
                                   + (" [BASELINE configs[4], the multi-GPU scaling frame]" if c5 else ""),
                       "tiles": "8x8, sharded across ranks in interleaved blocks of tile rows (~16 blocks per rank) by shm_shard_tiles" if world > 1 else "8x8",
                       "rays_per_step": rays / args.steps, "paths_per_step": tot["paths"] / args.steps,
                       "film_gather": "RCCL ncclSend/ncclRecv of each rank's film rows into rank 0's device film, inside the library and inside the "
                                      "timed region" if use_dist else "none"},
            # The primary roofline is the ceiling that BINDS the dominant kernel: VALU issue (lane-operations per second against what the
            # SIMDs can issue; frac <= 1 by construction). SURVEY §8(d)'s algorithmic-bytes figure — which counts every node visit as 32 B
            # from memory although four in five are cache hits, and therefore exceeds the HBM peak — is kept as roofline_hbm_algorithmic;
            # what HBM really carries (PMC traffic / launch time) is roofline_hbm_counter.
            "roofline": {"bound": "valu-issue", "kernel": "k_trace<closest> (BvhAggregate::intersect)",
                         "achieved": (valu["achieved_lane_ops_per_s"] / 1e12) if valu else None, "peak": (valu["peak_lane_ops_per_s"] / 1e12) if valu else None,
                         "unit": "Tlane-op/s", "frac": valu["frac"] if valu else None,
                         "frac_at_counter_pricing_4_clocks": valu["frac_at_counter_pricing_4_clocks"] if valu else None,
                         "frac_at_guide_peak_2_clocks": valu["frac_at_guide_peak_2_clocks"] if valu else None,
                         "traffic": traffic, "traffic_source": counters_src, "lanes_active": lanes,
                         "wait_frac": valu["wait_frac"] if valu else None, "salu_busy_frac": valu["salu_busy_frac_at_4_clocks"] if valu else None,
                         "valu_busy_frac": valu["valu_busy_frac_at_4_clocks"] if valu else None,
                         "avg_launch_ms": ms / launches, "launches": launches,
                         "bound_note": "lane-operations per second of the closest-hit traversal kernel over 1024 SIMDs x 64 lanes x clock / 2.4 clocks per "
                                       "instruction (the measured best issue rate; frac_at_guide_peak_2_clocks prices the same numerator against the guide's 2-cycle v_fma_f32: "
                                       "MI355X_MICROARCH.md, per-instruction cycle constants); the gathers are L2 / MALL-served (roofline_hbm_counter), so HBM is not "
                                       "the ceiling; see roofline_valu for every term" if valu else
                                       "no hardware counters available in this run (rocprofv3 missing and no committed profile of this configuration): "
                                       "see roofline_hbm_algorithmic"},
            "roofline_hbm_algorithmic": {"bound": "hbm (SURVEY 8d ALGORITHMIC bytes; cache-served, saturated: may exceed 1)",
                                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                         "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS, "bytes_per_launch": bytes_alg / launches,
                                         "nodes_per_ray": acc["nodes_closest"] / max(1, acc["rays_closest"]),
                                         "prims_per_ray": acc["tris_closest"] / max(1, acc["rays_closest"]),
                                         "closest_Mray_s_in_kernel": acc["rays_closest"] / (ms * 1e-3) / 1e6 if ms > 0 else 0.0},
            "roofline_hbm_counter": {"bound": "hbm (PMC traffic: FETCH_FACTOR x FETCH_SIZE + WRITE_SIZE per launch / launch time)", "traffic": traffic,
                                     "fetch_factor": FETCH_FACTOR, "achieved": hbm_counter, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": (hbm_counter / HBM_PEAK_GBS) if hbm_counter else None,
                                     "traffic_over_algorithmic": (traffic / (bytes_alg / launches)) if traffic else None},
            # the second traversal kernel, same accounting (informational; `roofline` above is the dominant kernel)
            "roofline_any_hit": {"kernel": "k_trace<any> (BvhAggregate::intersect_predicate)",
                                 "achieved": ((32.0 * acc["nodes_any"] + 48.0 * acc["tris_any"] + 48.0 * acc["rays_any"]) / (acc["ms_trace_any"] * 1e-3) / 1e9)
                                 if acc["ms_trace_any"] > 0 else 0.0, "unit": "GB/s",
                                 "nodes_per_ray": acc["nodes_any"] / max(1, acc["rays_any"]), "prims_per_ray": acc["tris_any"] / max(1, acc["rays_any"]),
                                 "any_Mray_s_in_kernel": acc["rays_any"] / (acc["ms_trace_any"] * 1e-3) / 1e6 if acc["ms_trace_any"] > 0 else 0.0},
            "breakdown_ms_per_step": {"trace_closest": acc["ms_trace_closest"] / args.steps, "trace_any": acc["ms_trace_any"] / args.steps,
                                      "shade_generate_film": acc["ms_shade"] / args.steps, "gpu_total": acc["ms_total"] / args.steps,
                                      "film_gather": acc.get("ms_gather", 0.0) / args.steps},
            "runtime": {"launcher": os.environ.get("SHM_LAUNCHED_BY") or ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "direct"),
                        "control_plane": "file store (128-byte id) + the library's RCCL collectives (barrier, max clock, counters)" if use_dist else "none",
                        "torch_imported": "torch" in sys.modules, "librccl": info["librccl_path"], "rccl_version": info["rccl_version"],
                        "libamdhip64": info["libamdhip_path"], "hip_runtime_version": info["hip_runtime_version"],
                        "rccl_ranks": info["rccl_ranks"], "rows_per_shard_block": info["rows_per_block"]},
        }
        if counters and counters.get("_shade"):
            sh = counters["_shade"]
            st, _, sv = counter_blocks(sh, sh.get("_duration_ms"))
            if sv:
                sv["kernel"] = "k_shade<false, true, false, true> (the fused vertex kernel of all-diffuse triangle scenes: integrator.rs:772-892)"
                out["roofline_shade"] = {"bound": "valu-issue at three waves per SIMD (informational: the fused vertex kernel, second to the closest-hit traversal in the frame)", "kernel": sv["kernel"],
                                         "avg_launch_ms_under_the_counter_pass": sh.get("_duration_ms"), "launches_per_step": sh.get("dispatches"),
                                         "achieved": sv["achieved_lane_ops_per_s"] / 1e12, "peak": sv["peak_lane_ops_per_s"] / 1e12, "unit": "Tlane-op/s", "frac": sv["frac"],
                                         "frac_at_counter_pricing_4_clocks": sv["frac_at_counter_pricing_4_clocks"], "lanes_active": sv["lanes_per_valu_inst"],
                                         "valu_busy_frac": sv["valu_busy_frac_at_4_clocks"], "wait_frac": sv["wait_frac"], "traffic": st,
                                         "hbm_counter_GBs": (st / (sh["_duration_ms"] * 1e-3) / 1e9) if st and sh.get("_duration_ms") else None}
        if valu:
            out["roofline_valu"] = valu
            if not (0.0 < valu["frac"] <= 1.0):
                # counters from the committed fallback profile of another build: reporting only — say so in the line, keep the measured run. LIVE counters that do not give
                # a fraction of the ceiling mean the accounting is broken: the line is printed and the run exits non-zero (below)
                frac_fatal = counters_src.startswith("live")
                out["roofline"]["frac_invalid"] = True
                out["roofline"]["frac_invalid_note"] = f"frac = {valu['frac']} is not a fraction of a ceiling: counters ({counters_src}) and this run's launch time do not belong together"
                out["roofline"]["frac"] = None
        if per_rank is not None:
            out["per_rank"] = per_rank
    if use_dist and rank == 0:
        # the gathered film: every pixel of the frame received all its samples, from some rank
        host = r.read_film()
        if not (host["weight_sum"] == float(args.spp)).all():
            raise SystemExit("gathered film is incomplete")
        out["nonfinite_pixels"] = int((~np.isfinite(host["rgb_sum"]).all(axis=-1)).sum())  # (reference behaviour in coated scenes: DESIGN.md §2)
        if out["nonfinite_pixels"] and not args.coated:
            raise SystemExit("gathered film is not finite")
        if world == 1:
            r.dist_selftest()  # film rows through the RCCL send / recv group, looped back to this rank
    if rank == 0 and world == 1 and not use_dist and not args.shard_of:
        # self-check outside the timed region: every pixel received all its samples, all sums finite
        t0 = time.perf_counter()
        host = r.read_film()
        t_readback = time.perf_counter() - t0
        # what the boundary moves over PCIe, never part of `value`: the scene goes up once at shm_scene_create, the film (32 B per pixel) comes back once per frame
        film_bytes = int(host.nbytes)
        out["host_transfers"] = {"scene_create_s": t_upload, "film_readback_ms": t_readback * 1e3, "film_bytes": film_bytes,
                                 "value_with_film_readback": out["value"] * (out["ms_per_step"] / (out["ms_per_step"] + t_readback * 1e3)),
                                 "note": "value counts the render with the scene resident in HBM; shm_scene_create (flatten + pair layout + H2D) is paid once per scene, the film read-back once per frame"}
        if not (host["weight_sum"] == float(args.spp)).all():
            raise SystemExit("film is incomplete")
        # LayeredBxDF::pdf of the reference can return 0 / 0 ("Reference quirks preserved"): counted, and fatal only where no
        # coated material exists
        out["nonfinite_pixels"] = int((~np.isfinite(host["rgb_sum"]).all(axis=-1)).sum())
        if out["nonfinite_pixels"] and not args.coated:
            raise SystemExit("film is not finite")
    if use_dist:
        r.dist_barrier()   # nobody tears its communicator down while a peer still checks the film
    r.close()
    if rank == 0 and world == 1:
        # the side results BEFORE the CPU baseline (round 6): after the oracle's run on every host thread the box's host side stays unsettled for a minute or more and
        # launch-bound frames pay for it — C4's second timed frame 373-381 ms instead of 347, every other side result 0.5-1.5 % slower (profiles/r06_bench_order.txt)
        if not args.no_side and not args.coated and not args.variant and not args.shard_of and (width, height) == (1024, 1024):
            out["side_results"] = side_results(lib, args, render, scenes, sc, log)
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sc, params, lib)
            except Exception as e:  # the baseline is reporting only; never let it hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Mray/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    if store is not None:
        store.finish()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is flushed at exit: flush it now so that the JSON line is the
        # last thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
        if frac_fatal:
            raise SystemExit("roofline.frac is not a fraction of its ceiling although the counters are this run's own: the accounting is broken (line printed above)")


def dry_run(args, rank, local_rank, world, store):
    """The control plane without a GPU (CPU tests): environment, id exchange, barrier and max-reduce — through the file store, since no
    communicator can exist here. Rank 0 prints one JSON line."""
    fake_id = store.broadcast("rccl_unique_id", bytes((7 * i + 1) & 0xFF for i in range(128)) if rank == 0 else None)
    if rank == args.dry_run_fail_rank:
        log(f"[bench] dry run: rank {rank} fails on request")
        raise SystemExit(7)
    store.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    store.barrier()
    dt = store.allreduce_max("dt", time.perf_counter() - t0)
    ranks = store.allgather("who", json.dumps({"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "ppid": os.getppid(),
                                               "torch_imported": "torch" in sys.modules, "id_ok": fake_id[:3] == bytes((1, 8, 15))}).encode())
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "max_dt_s": dt, "ranks": [json.loads(x) for x in ranks],
                          "launcher": os.environ.get("SHM_LAUNCHED_BY") or "external"}), flush=True)
    store.finish()


if __name__ == "__main__":
    main()
