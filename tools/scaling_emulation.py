#!/usr/bin/env python3
"""Predicted strong scaling of a BASELINE.json configuration from ONE GPU (GPU box): `bench.py --emulate-world N --emulate-rank all` renders
EVERY rank's share of an N-way split in turn, N = 1, 2, 4, 8, with both partitions of SURVEY section 8(e); a step of the N-GPU job takes as long as
its slowest rank, plus the all-reduce of the fp32 film, added as a ring estimate over xGMI (per-link bound: 2 (N-1)/N x bytes / 153 GB/s; the
driver's SCALE run would measure the real one -- NO hardware curve exists yet).  Writes gpurun_out/r04_scaling_emulation_<config>.json.
usage: tools/scaling_emulation.py [config=c2] [steps=2] [shards=samples,tiles]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = sys.argv[2] if len(sys.argv) > 2 else "2"
shards = (sys.argv[3] if len(sys.argv) > 3 else "samples,tiles").split(",")
XGMI_LINK_GBS = 153.0
sys.path.insert(0, ROOT)
import bench as _bench
# the job that is split: the config's own spp where a step is the whole job (c2: 1024), else a slice of it 8 x the bench's step (c4: 1024 of 4096 spp,
# c5: 256 of 16384) -- so that an 8-way split leaves every rank the bench's own step, not an eighth of a step that was already a reduced sample
_cfg = _bench.CONFIGS[cfg]
job_spp = min(_cfg["spp"], 8 * _cfg["step_spp"])
out = {"config": cfg, "job_spp": None, "method": __doc__.split("usage")[0].strip(), "rows": []}
out["job_spp"] = job_spp
base = {}
for shard in shards:
    for n in (1, 2, 4, 8):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", steps, "--warmup", "1", "--no-cpu-baseline", "--shard", shard, "--spp", str(job_spp)]
        if n > 1: cmd += ["--emulate-world", str(n), "--emulate-rank", "all"]
        line = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
        d = json.loads(line)
        W, H = (800, 800) if cfg == "c2" else (1280, 720) if cfg == "c3" else (1920, 1080) if cfg == "c4" else (3840, 2160)
        film = W * H * 16
        allreduce_ms = 0.0 if n == 1 else 2.0 * (n - 1) / n * film / (XGMI_LINK_GBS * 1e9) * 1e3
        ms = d["ms_per_step"]
        if n == 1: base[shard] = ms
        per_rank = [r["ms_per_step"] for r in (d.get("emulated_ranks") or [])]
        row = {"shard": shard, "world": n, "rank_ms": ms, "rank_ms_min": min(per_rank) if per_rank else ms, "rank_ms_all": per_rank or [ms],
               "rank_spread": round((max(per_rank) - min(per_rank)) / max(per_rank), 4) if per_rank else 0.0, "allreduce_ms_estimate": round(allreduce_ms, 3), "job_ms": round(ms + allreduce_ms, 3),
               "efficiency": round(base[shard] / (n * (ms + allreduce_ms)), 4), "mray_per_s_per_gpu": d["value"], "workload": d["config"]["workload"],
               "film_count_plane_ok": d["film_count_plane_ok"]}
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_scaling_emulation_%s.json" % cfg), "w"), indent=1)
