"""Per-dispatch rocprofv3 counters joined with the kernel trace of the same pass, averaged per kernel (optionally per duration
bucket): shader clock, MFMA busy, VALU busy, wave-cycle split, VALU instructions per score element for the attention kernels.

  MFMA busy %  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles of the dispatch); shader cycles = GRBM_GUI_ACTIVE / 8
                 (rocprofv3 sums the counter over the 8 XCDs); shader clock = cycles / wall time (MI355X_MICROARCH.md, DVFS)
  VALU busy %  = 4 x SQ_ACTIVE_INST_VALU / (1024 x shader cycles)   (SQ_ACTIVE_INST_* count quad-cycles)
  waves / SIMD = SQ_WAVE_CYCLES x 4 / (1024 x shader cycles)        (time-averaged resident waves)
  VALU / score = SQ_INSTS_VALU x 64 lanes / executed score elements (--scores N: b x heads x visited 64 x 64 blocks x 4096)

Usage: pmc_kernels_report.py --match attn [--scores 6.0e8] <dirA> [<dirB> ...]   (separate --pmc passes: one directory each)"""
import argparse, collections, csv, glob, re


def load(d):
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"])
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        per[r["Dispatch_Id"]]["_vgpr"] = float(r.get("VGPR_Count", 0) or 0) + float(r.get("Accum_VGPR_Count", 0) or 0)
    return dur, per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--match", default="")
    ap.add_argument("--scores", type=float, default=0.0)
    ap.add_argument("dirs", nargs="+")
    a = ap.parse_args()
    for d in a.dirs:
        dur, per = load(d)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for did, c in per.items():
            ns, name = dur.get(did, (0.0, "?"))
            if a.match not in name:
                continue
            key = re.sub(r"_ZN12_GLOBAL__N_1\d+|\(anonymous namespace\)::", "", name)[:64]
            for k, v in c.items():
                agg[key][k].append(v)
            agg[key]["ns"].append(ns)
        print("== " + d)
        for key, c in sorted(agg.items()):
            m = {k: sum(v) / len(v) for k, v in c.items()}
            line = f"{key:66s} n={len(c['ns']):3d} {m['ns'] / 1e3:8.1f} us  regs {m.get('_vgpr', 0):.0f}"
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0 if "GRBM_GUI_ACTIVE" in m else None
            if cyc:
                line += f"  shader clock {cyc / m['ns']:5.3f} GHz"
                if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                    line += f"  MFMA busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):5.1f} %"
                if "SQ_ACTIVE_INST_VALU" in m:
                    line += f"  VALU busy {100 * 4 * m['SQ_ACTIVE_INST_VALU'] / (1024.0 * cyc):5.1f} %"
                if "SQ_WAVE_CYCLES" in m:
                    line += f"  waves/SIMD {4 * m['SQ_WAVE_CYCLES'] / (1024.0 * cyc):4.2f}"
            if "SQ_INSTS_VALU" in m:
                line += f"  VALU insts {m['SQ_INSTS_VALU']:.3g}  MFMA insts {m.get('SQ_INSTS_MFMA', 0):.3g}  LDS insts {m.get('SQ_INSTS_LDS', 0):.3g}"
                if a.scores:
                    line += f"  VALU/score {m['SQ_INSTS_VALU'] * 64 / a.scores:5.2f}"
            if "SQ_WAIT_ANY" in m:
                wc = m.get("SQ_ACTIVE_INST_ANY", 0) + m.get("SQ_WAIT_ANY", 0) + m.get("SQ_WAIT_INST_ANY", 0)
                line += "  wave cycles: issuing %.0f %% parked %.0f %% waiting-to-issue %.0f %% (lds %.0f %%)" % (
                    100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc,
                    100 * m.get("SQ_WAIT_INST_LDS", 0) / wc)
            print(line)


if __name__ == "__main__":
    main()
