#!/usr/bin/env python3
"""Collect hardware counters for the hot kernels of bench.py, one `rocprofv3 --pmc` pass per counter group
(counters are never split over passes by the tool, and a pass must stay within the per-block limits), and
write per-kernel averages per launch to gpurun_out/<tag>_pmc.json and gpurun_out/traffic.json (copied into
profiles/ afterwards: only gpurun_out/ travels back from the GPU box).

    python3 profiles/collect_pmc.py <tag> [passes]   # on the GPU box, from the repository root; passes e.g. 0,1

Every pass is its own short bench run (1 warm-up + 1 timed volume, no CPU baseline) under a hard time limit;
the script stops at the first pass that fails.  HBM bytes: FETCH_SIZE and WRITE_SIZE are reported in KiB; on
gfx950 FETCH_SIZE counts wide coalesced reads at half their size, so read bytes = 2 * FETCH_SIZE * 1024.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = "k_integrate|k_front|k_interp|k_prep|k_gate_reduce|k_geom_pts"
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
     "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_LDS"],
    ["SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INSTS_VALU_TRANS_F32",
     "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_INSTS_LDS"],
    ["TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_ADDR_STALL_CYCLES_sum",
     "TA_FLAT_READ_WAVEFRONTS_sum", "TA_TA_BUSY_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]


def run_pass(counters, outdir):
    if os.path.isdir(outdir):
        shutil.rmtree(outdir)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-include-regex", KERNELS, "--output-format", "csv",
                                               "-d", outdir, "-o", "pmc", "--",
                                               "python3", "bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-host-copy"]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd=ROOT, env=env, check=True, timeout=240, stdout=subprocess.DEVNULL,
                   stderr=open(os.path.join(ROOT, "gpurun_out", "pmc_last.err"), "w"))
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise RuntimeError("no counter_collection.csv under " + outdir)
    acc = {}
    with open(files[0]) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            d = acc.setdefault(k, {}).setdefault(row["Counter_Name"], {})
            key = row.get("Dispatch_Id", row.get("Correlation_Id"))
            d[key] = d.get(key, 0.) + float(row["Counter_Value"])
    return {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} | {"_launches": max(len(v) for v in cs.values())}
            for k, cs in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    result = {"command": "rocprofv3 --pmc <group> --kernel-include-regex \"%s\" -- python3 bench.py --steps 1 --warmup 1 "
                         "--no-cpu-baseline --no-host-copy" % KERNELS,
              "note": "one pass per counter group; values are averages per launch", "kernels": {}}
    only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else range(len(PASSES))
    for i, group in enumerate(PASSES):
        if i not in only:
            continue
        print("pass %d: %s" % (i, " ".join(group)), flush=True)
        res = run_pass(group, os.path.join(ROOT, "gpurun_out", "pmc_pass%d" % i))
        for k, cs in res.items():
            result["kernels"].setdefault(k, {}).update(cs)
    for k, cs in result["kernels"].items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            cs["hbm_bytes_per_launch"] = (2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.
    with open(os.path.join(ROOT, "gpurun_out", tag + "_pmc.json"), "w") as fh:
        json.dump(result, fh, indent=1, sort_keys=True)
    integ = [v for k, v in result["kernels"].items() if k.startswith("k_integrate<")]
    if integ and "hbm_bytes_per_launch" in integ[0]:
        t = {"k_integrate_hbm_bytes_per_launch": integ[0]["hbm_bytes_per_launch"],
             "source": "profiles/%s_pmc.json (rocprofv3 --pmc passes of profiles/collect_pmc.py; 2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)" % tag}
        fr = result["kernels"].get("k_front", {})
        if "SQ_INSTS_VALU" in fr:
            t["k_front_valu_insts_per_launch"] = fr["SQ_INSTS_VALU"]
        if "TA_TA_BUSY_sum" in fr and "SQ_BUSY_CYCLES" in fr:
            # TA_TA_BUSY_sum adds the 256 CUs' busy cycles, SQ_BUSY_CYCLES the 32 shader engines'
            t["k_front_ta_busy_frac"] = (fr["TA_TA_BUSY_sum"] / 256.) / (fr["SQ_BUSY_CYCLES"] / 32.)
        gr = next((v for k, v in result["kernels"].items() if k.startswith("k_gate_reduce")), {})
        if "hbm_bytes_per_launch" in gr:
            t["k_gate_reduce_hbm_bytes_per_launch"] = gr["hbm_bytes_per_launch"]
        with open(os.path.join(ROOT, "gpurun_out", "traffic.json"), "w") as fh:
            json.dump(t, fh, indent=1)
    print("done")


if __name__ == "__main__":
    main()
