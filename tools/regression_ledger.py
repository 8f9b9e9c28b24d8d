"""Regression ledger across ALL configs: every number a round's artifacts carry, diffed against the
previous round's artifact of the same name, every change beyond +-2 % flagged.

    python tools/regression_ledger.py r06z r05z [--threshold 2.0] > profiles/r06_vs_r05.txt

Reads profiles/<tag>_* (what tools/profile_round.sh and tools/record_artifacts.sh wrote and the builder copied):
  *_bench.json-like files          -> value, config.single_update_steps_per_s, roofline.frac, dominant_kernel in-step rate
  *_online_bench.txt               -> update-steps/s per --env-us
  *_metrics_on_bench.txt           -> update-steps/s
  *_act_bench.txt                  -> us per call (lower is better)
  *_soak_100k.txt                  -> updates/s
  *_kernel_stats.txt               -> per-kernel us per update of the kernels above 10 us (lower is better)
A line is only compared when both rounds have it; what exists on one side only is listed at the end.
"""
from __future__ import annotations

import argparse
import json
import re
import sys
from pathlib import Path

PROFILES = Path(__file__).resolve().parents[1] / "profiles"


def _json_lines(path: Path):
    for line in path.read_text(errors="replace").splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                yield json.loads(line)
            except json.JSONDecodeError:
                pass


def extract(tag: str) -> dict:
    """{label: (value, higher_is_better)}"""
    out: dict = {}
    for path in sorted(PROFILES.glob(f"{tag}_*")):
        if not path.is_file():
            continue
        name = path.name[len(tag) + 1:]
        stem = name.rsplit(".", 1)[0]
        text = path.read_text(errors="replace") if path.stat().st_size < 4 << 20 else ""
        if name.endswith((".json", ".log")) and "pmc" not in name:
            for d in _json_lines(path):
                if not isinstance(d, dict) or "value" not in d:
                    continue
                out[f"{stem}: value [{d.get('unit', '')}]"] = (float(d["value"]), bool(d.get("higher_is_better", True)))
                cfg = d.get("config") or {}
                if isinstance(cfg, dict) and cfg.get("single_update_steps_per_s"):
                    out[f"{stem}: single_update_steps_per_s"] = (float(cfg["single_update_steps_per_s"]), True)
                roof = d.get("roofline") or {}
                if roof.get("frac") is not None:
                    out[f"{stem}: roofline.frac"] = (float(roof["frac"]), True)
                dk = roof.get("dominant_kernel") or {}
                if isinstance(dk.get("in_step"), dict) and dk["in_step"].get("frac_of_peak") is not None:
                    out[f"{stem}: dominant kernel in-step frac"] = (float(dk["in_step"]["frac_of_peak"]), True)
                cb = d.get("cpu_baseline") or {}
                if cb.get("value") is not None:
                    out[f"{stem}: cpu_baseline [{cb.get('cores')} threads]"] = (float(cb["value"]), True)
        elif name.endswith("online_bench.txt"):
            for m in re.finditer(r"synthetic env \(([\d.]+) us/step\).*?-> (\d+) env-steps/s, (\d+) update-steps/s", text):
                out[f"online loop, env {float(m.group(1)):g} us: update-steps/s"] = (float(m.group(3)), True)
                out[f"online loop, env {float(m.group(1)):g} us: env-steps/s"] = (float(m.group(2)), True)
        elif name.endswith("metrics_on_bench.txt"):
            m = re.search(r"([\d.]+) update-steps/s", text)
            if m:
                out["metrics ON (use_tb=1): update-steps/s"] = (float(m.group(1)), True)
        elif name.endswith("act_bench.txt"):
            for m in re.finditer(r"^\s*(.+?):\s+([\d.]+) us per call", text, re.M):
                out[f"batch-1 latency {m.group(1).strip()} [us]"] = (float(m.group(2)), False)
        elif name.endswith("soak_100k.txt"):
            m = re.search(r"\((\d+)/s\)", text)
            if m:
                out["soak 100k updates: updates/s"] = (float(m.group(1)), True)
        elif name.endswith("kernel_stats.txt"):
            # tools/prof_summary.py table: "<name> <calls> <total us> <avg us> <min> <max> <pct>"; us per update needs the update
            # count: use the per-launch average of the kernels above 1 % instead (independent of how many steps the trace held)
            for line in text.splitlines():
                m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
                if m and float(m.group(7)) >= 1.0:
                    kname = re.sub(r"\(.*$", "", m.group(1).replace("void ", "").replace("fbhip::", "").replace("(anonymous namespace)::", "")).strip()
                    out[f"{stem}: avg us per launch, {kname}"] = (float(m.group(4)), False)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("new")
    ap.add_argument("old")
    ap.add_argument("--threshold", type=float, default=2.0, help="percent")
    a = ap.parse_args()
    new, old = extract(a.new), extract(a.old)
    both = [k for k in new if k in old]
    print(f"# regression ledger: profiles/{a.new}_* against profiles/{a.old}_*  (threshold +-{a.threshold:g} %; "
          f"{len(both)} lines compared)")
    flagged = 0
    rows = []
    for k in both:
        (v1, hib), (v0, _) = new[k], old[k]
        if v0 == 0:
            continue
        pct = 100.0 * (v1 - v0) / abs(v0)
        better = (pct > 0) == hib
        mark = ""
        if abs(pct) > a.threshold:
            mark = "BETTER" if better else "WORSE"
            flagged += 1
        rows.append((mark, k, v0, v1, pct))
    for mark, k, v0, v1, pct in sorted(rows, key=lambda r: (r[0] != "WORSE", r[0] != "BETTER", r[1])):
        print(f"{mark:7s} {pct:+7.1f} %  {v0:12.4g} -> {v1:12.4g}  {k}")
    print(f"# {flagged} lines beyond +-{a.threshold:g} %")
    only_new, only_old = [k for k in new if k not in old], [k for k in old if k not in new]
    if only_new:
        print(f"# only in {a.new}: " + "; ".join(f"{k} = {new[k][0]:.4g}" for k in only_new))
    if only_old:
        print(f"# only in {a.old}: " + "; ".join(only_old))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
