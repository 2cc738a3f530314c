"""Per-dispatch counters of gemm_w4_kernel joined with the kernel trace (durations) of the same rocprofv3 pass.
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x shader cycles of the dispatch); shader cycles = GRBM_GUI_ACTIVE / 8
(rocprofv3 sums the counter over the 8 XCDs: a 72-us dispatch reads 1.38e6 = 8 x 72 us x 2.39 GHz); shader clock = cycles /
wall time (MI355X_MICROARCH.md "DVFS give-back").  Usage: pmc_gemm_report.py <dirA> [<dirB>]"""
import collections, csv, glob, re, sys

PEAK_FLOPS_PER_CU_CLK = 2.5e15 / 256 / 2.4e9          # dense bf16 / fp16 MFMA peak per CU and shader cycle (2.4 GHz nominal)


def load(d):
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"])
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        per[r["Dispatch_Id"]]["_grid"] = r.get("Grid_Size", "")
    return dur, per


def main():
    for d in sys.argv[1:]:
        dur, per = load(d)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for did, c in per.items():
            ns, name = dur.get(did, (0.0, "?"))
            if "gemm_w4" not in name:
                continue
            key = re.sub(r"_ZN12_GLOBAL__N_1\d+|\(anonymous namespace\)::", "", name)[:60] + " ~%dus" % (round(ns / 1e3 / 50) * 50)
            for k, v in c.items():
                if k != "_grid":
                    agg[key][k].append(v)
            agg[key]["ns"].append(ns)
        print("== " + d)
        for key, c in sorted(agg.items()):
            m = {k: sum(v) / len(v) for k, v in c.items()}
            line = f"{key:78s} n={len(c['ns']):3d} {m['ns'] / 1e3:8.1f} us"
            if "GRBM_GUI_ACTIVE" in m:
                cyc = m["GRBM_GUI_ACTIVE"] / 8.0                         # the counter is summed over the 8 XCDs
                clk = cyc / m["ns"]                                      # cycles per ns = GHz
                busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)      # 1024 SIMDs; 16 busy cycles per 16x16x32 MFMA
                line += f"  shader clock {clk:5.3f} GHz  MFMA busy {100 * busy:5.1f} % of SIMD cycles  MFMA insts {m.get('SQ_INSTS_MFMA', 0):.3g}"
                line += f"  -> peak at this clock {2500.0 * clk / 2.4:6.0f} TFLOP/s"
            if "SQ_WAIT_ANY" in m:
                wc = m.get("SQ_ACTIVE_INST_ANY", 0) + m.get("SQ_WAIT_ANY", 0) + m.get("SQ_WAIT_INST_ANY", 0)
                line += "  wave cycles: active %.0f %% wait_any %.0f %% wait_inst %.0f %% (lds %.0f %%)" % (
                    100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc,
                    100 * m.get("SQ_WAIT_INST_LDS", 0) / wc)
            print(line)


if __name__ == "__main__":
    main()
