"""profiles/traffic.json (what bench.py's roofline.traffic quotes without --pmc) from the two PMC summaries of a round:
    python tools/make_traffic_json.py r06z [quadruped]
reads profiles/<tag>[_quadruped]_pmc_FETCH_SIZE.txt / _pmc_WRITE_SIZE.txt (tools/pmc_summary.py), applies the guide's gfx950
correction (FETCH_SIZE x 2: wide coalesced reads are tallied at half, MI355X_MICROARCH.md "HBM") and ties the record to the digest
of controllable_agent_amd/csrc (bench.kernel_sources_sha16) so the bench line can say whether it belongs to the build it ran."""
import json, re, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench

tag = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "walker"
pre = f"{tag}_" + ("" if wl == "walker" else f"{wl}_")


def total(counter):
    txt = (ROOT / "profiles" / f"{pre}pmc_{counter}.txt").read_text()
    return float(re.search(r"TOTAL per update step\s+([\d.]+)", txt).group(1))


f, w = total("FETCH_SIZE"), total("WRITE_SIZE")
rec = {"round": tag, "fetch_bytes_per_update_raw": f, "fetch_bytes_per_update_corrected": 2 * f, "write_bytes_per_update": w,
       "hbm_bytes_per_update": 2 * f + w,
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of a short bench.py run, summed over "
               "every kernel of one update (tools/pmc_summary.py); FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM' (gfx950 wide coalesced "
               "reads are tallied at half); WRITE_SIZE uncalibrated; Infinity-Cache hits are counted in both",
       "sources": [f"profiles/{pre}pmc_FETCH_SIZE.txt", f"profiles/{pre}pmc_WRITE_SIZE.txt"],
       "kernel_sources_sha16": bench.kernel_sources_sha16()}
out = ROOT / "profiles" / ("traffic.json" if wl == "walker" else f"traffic_{wl}.json")
out.write_text(json.dumps(rec, indent=1) + "\n")
print(out, rec["hbm_bytes_per_update"] / 1e9, "GB per update")

if wl == "walker":
    # the in-step GEMM record of the same round (tools/profile_round.sh wrote profiles/<tag>_dominant_in_step.json): tie it to the build too
    src = ROOT / "profiles" / f"{tag}_dominant_in_step.json"
    if src.exists():
        d = json.loads(src.read_text())
        d["tag"], d["kernel_sources_sha16"] = tag, bench.kernel_sources_sha16()
        (ROOT / "profiles" / "dominant_in_step.json").write_text(json.dumps(d, indent=1) + "\n")
        print("profiles/dominant_in_step.json <-", src.name)
