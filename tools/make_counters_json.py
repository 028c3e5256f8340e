"""Condense the per-kernel counter CSVs written by tools/pmc_bench.sh into the JSON that
bench.py reads (HBM traffic per launch, VALU issue utilisation).

  python tools/make_counters_json.py <tag> <grid> <out.json>     reads gpurun_out/pmcb_<tag>_{1..4}.csv
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import lib_source_hash  # noqa: E402

SLOTS = {  # profiler slot of the library -> kernel name prefix in the rocprofv3 tables
    "fft_z_conv": "void k_fft_z_conv<",
    "fft_y_fwd": "void k_fft_y_fwd<",
    "fft_y_inv": "void k_fft_y_inv<",
    "fft_x_fwd": "void k_fft_x_fwd<",
    "fft_x_inv": "void k_fft_x_inv<",
    "fused_advect_diffuse": "void k_advect_diffuse<",
    "fused_advect": "void k_advect_tiled<",
    "fused_diffuse": "void k_diffuse_tiled<",
    "curl": "void k_curl_tiled<",
}
# scalar f32 VALU issue rate with packed math off, measured by tools/micro/valu_rate.hip at four
# or more waves per SIMD: one wave64 instruction per 1.79 ns per SIMD, 1024 SIMDs
VALU_WAVE_INSTR_PER_S = 1024 / 1.79e-9


def main():
    tag, grid, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    d = collections.defaultdict(dict)
    for i in (1, 2, 3, 4):
        try:
            rows = list(csv.DictReader(open(f"gpurun_out/pmcb_{tag}_{i}.csv")))
        except OSError:
            continue
        for r in rows:
            d[r["kernel"]][r["counter"]] = float(r["avg_per_dispatch"])
    kernels = {}
    for slot, prefix in SLOTS.items():
        names = [k for k in d if k.startswith(prefix)]
        if not names:
            continue
        # a slot may cover several launches (the y passes: 16-column tiles, then narrow tiles for the column left
        # over): additive counters are summed over them, the wave-time split is that of the longest launch
        names.sort(key=lambda k: -d[k].get("SQ_WAVE_CYCLES", d[k].get("FETCH_SIZE", 0.0)))
        c = dict(d[names[0]])
        for extra in names[1:]:
            for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
                if counter in c and counter in d[extra]:
                    c[counter] += d[extra][counter]
        rec = {"kernel": names[0]}
        if len(names) > 1:
            rec["also"] = names[1:]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["fetch_size_kib_raw"] = c["FETCH_SIZE"]
            rec["write_size_kib"] = c["WRITE_SIZE"]
            rec["hbm_bytes"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            rec["valu_wave_instructions"] = c["SQ_INSTS_VALU"]
            rec["valu_floor_ms"] = c["SQ_INSTS_VALU"] / VALU_WAVE_INSTR_PER_S * 1e3
            rec["shader_cycles"] = c["GRBM_GUI_ACTIVE"] / 8
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            rec["wave_time_split"] = {k: round(c[n] / wc, 3) for k, n in (
                ("parked_waitcnt_or_barrier", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"),
                ("issuing", "SQ_ACTIVE_INST_ANY"), ("issuing_valu", "SQ_ACTIVE_INST_VALU"),
                ("issuing_lds", "SQ_ACTIVE_INST_LDS")) if n in c}
        kernels[slot] = rec
    rec = {
        "_about": "Per-launch counters of each hot kernel in the f32 navier_stokes step on one MI355X, from separate "
                  "rocprofv3 --pmc passes over `python3 bench.py --steps 3 --warmup 1` (tools/pmc_bench.sh): FETCH_SIZE, "
                  "WRITE_SIZE (KiB per dispatch), SQ instruction and wave-cycle counters. hbm_bytes = 2 * FETCH_SIZE + "
                  "WRITE_SIZE: on gfx950 FETCH_SIZE tallies the 128-byte requests of coalesced streaming reads at 64 "
                  "bytes; the factor reproduces the known read volume of the FFT passes. Infinity-cache hits are "
                  "included in FETCH_SIZE. valu_floor_ms = wave-level VALU instructions / the scalar-f32 issue rate "
                  "of the chip with packed math off (one wave64 instruction per 1.79 ns per SIMD, "
                  "tools/micro/valu_rate.hip).",
        "grid": [grid] * 3,
        "dtype": "f32",
        # the counters describe THIS build of the kernels: bench.py withholds them when the sources have changed
        "lib_source_sha256": lib_source_hash(),
        "kernels": kernels,
    }
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    for k, v in kernels.items():
        print(k, {a: b for a, b in v.items() if a in ("hbm_bytes", "valu_floor_ms")})


if __name__ == "__main__":
    main()
