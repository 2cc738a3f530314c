import csv, collections, re, sys, glob
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for n, c in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in n: continue
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    short = re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+', '', n)[:70]
    print(short)
    print("   " + " ".join(f"{k}={v:.3g}" for k, v in sorted(m.items())))
    if 'SQ_LDS_IDX_ACTIVE' in m:
        print("   LDS conflict/active=%.2f unaligned=%.3g | wait_any %.0f%% wait_inst %.0f%% (lds %.0f%%) active %.0f%% of wave cycles" % (
            m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m['SQ_LDS_IDX_ACTIVE'], 1), m.get('SQ_LDS_UNALIGNED_STALL', 0), 100 * m.get('SQ_WAIT_ANY', 0) / wc,
            100 * m.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * m.get('SQ_WAIT_INST_LDS', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc))
