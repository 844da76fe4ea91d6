"""Developer tool: profiles/<label>.txt (summarize_rocprof.py output) -> the small JSON bench.py quotes (HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE PMC passes, LDS pipe figures).  usage: python tools/make_traffic_json.py profiles/r02_step_kernel_case14.txt 16"""
import json, re, sys
txt = open(sys.argv[1]).read()
steps_per_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
alg_bytes_per_launch = float(sys.argv[3]) if len(sys.argv) > 3 else None      # algorithmic bytes of one launch (bench.py: bytes/step x lanes x steps)
def ctr(name):
    m = re.search(rf"^\s*{name} =\s+([0-9.]+)\s+\(n=\d+, avg dispatch ([0-9.]+) us\)\s+(.*)$", txt, re.M)
    return (float(m.group(1)), float(m.group(2)), m.group(3).strip()) if m else (None, None, None)
fetch, _, kern = ctr("FETCH_SIZE")
write, _, _ = ctr("WRITE_SIZE")
lds_act, disp_us, _ = ctr("SQ_LDS_IDX_ACTIVE")
conf, _, _ = ctr("SQ_LDS_BANK_CONFLICT")
valu, _, _ = ctr("SQ_ACTIVE_INST_VALU")
wave_cyc, _, _ = ctr("SQ_WAVE_CYCLES")
waves, _, _ = ctr("SQ_WAVES")
geo = re.search(r"dispatches=(\d+) avg_ns=(\d+).*?lds_bytes=(\d+) vgpr=(\d+)", txt)
avg_us = int(geo.group(2)) / 1e3
clk_ghz = 2.4
out = {
    "source": sys.argv[1], "kernel": kern, "env_steps_per_launch": steps_per_launch,
    "kernel_trace_avg_us": avg_us, "avg_launch_us": avg_us,
    "algorithmic_bytes_per_launch": alg_bytes_per_launch,
    "traffic_over_algorithmic": ((2.0 * fetch + write) * 1024.0 / alg_bytes_per_launch) if alg_bytes_per_launch else None, "lds_bytes_per_block": int(geo.group(3)), "vgpr": int(geo.group(4)),
    "fetch_size_kib": fetch, "write_size_kib": write,
    # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read -> doubled; WRITE_SIZE as reported
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "hbm_bytes_per_env_step_batch": (2.0 * fetch + write) * 1024.0 / steps_per_launch,
    "lds_pipe_busy_frac": lds_act / 256.0 / (disp_us * 1e-6 * clk_ghz * 1e9),
    "lds_bank_conflict_frac_of_lds_cycles": conf / lds_act,
    "valu_busy_frac": 4.0 * valu / 1024.0 / (disp_us * 1e-6 * clk_ghz * 1e9),
    "waves_per_simd": waves / 1024.0,
    "note": "PMC passes of `python bench.py --steps 48 --warmup 16 --windows 1 --no-cpu-baseline --no-secondary` under rocprofv3 --pmc (one counter "
            "group per pass, tools/profile_round.sh); busy fractions assume 2.4 GHz (profiled runs clock lower: upper bounds)",
}
print(json.dumps(out, indent=1))
