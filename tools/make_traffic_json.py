"""Developer tool: profiles/<label>.txt (summarize_rocprof.py output) -> the small JSON bench.py quotes (HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE PMC passes, LDS pipe figures).  usage: python tools/make_traffic_json.py profiles/r02_step_kernel_case14.txt 16"""
import json, re, sys
txt = open(sys.argv[1]).read()
steps_per_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
alg_bytes_per_launch = float(sys.argv[3]) if len(sys.argv) > 3 else None      # algorithmic bytes of one launch (bench.py: bytes/step x lanes x steps)
kfilter = sys.argv[4] if len(sys.argv) > 4 else "step_sparse_kernel"            # substring of the kernel the JSON is about
def ctr(name):
    for m in re.finditer(rf"^\s*{name} =\s+([0-9.]+)\s+\(n=\d+, avg dispatch ([0-9.]+) us\)\s+(.*)$", txt, re.M):
        if kfilter in m.group(3):
            return float(m.group(1)), float(m.group(2)), m.group(3).strip()
    return None, None, None
fetch, _, kern = ctr("FETCH_SIZE")
write, _, _ = ctr("WRITE_SIZE")
lds_act, disp_us, _ = ctr("SQ_LDS_IDX_ACTIVE")
conf, _, _ = ctr("SQ_LDS_BANK_CONFLICT")
valu, _, _ = ctr("SQ_ACTIVE_INST_VALU")
wave_cyc, _, _ = ctr("SQ_WAVE_CYCLES")
waves, _, _ = ctr("SQ_WAVES")
geo = None
for mk in re.finditer(r"^(\S.*)\n\s+dispatches=(\d+) avg_ns=(\d+).*?lds_bytes=(\d+) vgpr=(\d+)", txt, re.M):
    if kfilter in mk.group(1):
        geo = mk
        break
geo = re.match(r"(\d+) (\d+) (\d+) (\d+)", " ".join(geo.group(k) for k in (2, 3, 4, 5)))
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
    "lds_pipe_busy_frac": lds_act / 256.0 / (disp_us * 1e-6 * clk_ghz * 1e9) if lds_act else None,
    "lds_bank_conflict_frac_of_lds_cycles": conf / lds_act if lds_act and conf is not None else None,
    "valu_busy_frac": 4.0 * valu / 1024.0 / (disp_us * 1e-6 * clk_ghz * 1e9) if valu else None,
    "waves_per_simd": waves / 1024.0 if waves else None,
    "note": "PMC passes of the bench workload under rocprofv3 --pmc (one counter group per pass, observation trajectory on in every dispatch: "
            "tools/profile_r05.sh); busy fractions assume 2.4 GHz (profiled runs clock lower: upper bounds)",
}
print(json.dumps(out, indent=1))
