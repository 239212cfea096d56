"""Histograms from a rocprofv3 PC-sampling run (csv output): samples of the dispatches whose kernel name matches a regex, grouped by source
line (line-table build), by instruction, by stall reason / instruction type (stochastic sampling).
    python tools/pcsamp_summary.py RAW_DIR KERNEL_REGEX OUT_DIR     (prints the summary; writes OUT_DIR/samples_*.csv.gz = the matching rows)"""
import collections
import csv
import glob
import gzip
import os
import re
import sys

raw, kre, out = sys.argv[1], re.compile(sys.argv[2]), sys.argv[3]
csv.field_size_limit(1 << 30)
ktrace = glob.glob(os.path.join(raw, "**", "*kernel_trace.csv"), recursive=True)
names = {}
for f in ktrace:
    for row in csv.DictReader(open(f)):
        names[row.get("Dispatch_Id")] = row.get("Kernel_Name", "?").split("(")[0]
files = [f for f in glob.glob(os.path.join(raw, "**", "*pc_sampling*.csv"), recursive=True)]
print("# files:", [os.path.basename(f) for f in files], " dispatches named:", len(names))
for f in files:
    rd = csv.DictReader(open(f))
    cols = rd.fieldnames
    print("# columns of %s: %s" % (os.path.basename(f), cols))
    tot = 0
    per_kernel = collections.Counter()
    by = {c: collections.Counter() for c in ("Instruction_Comment", "Instruction", "Stall_Reason", "Instruction_Type", "Wave_Issued_Instruction")}
    issued_by_line = collections.Counter()
    stall_by_line = collections.defaultdict(collections.Counter)
    keep = gzip.open(os.path.join(out, "samples_%s.csv.gz" % os.path.basename(f).split(".")[0][-24:]), "wt")
    w = csv.writer(keep); w.writerow(["Kernel"] + cols)
    kept = 0
    for row in rd:
        tot += 1
        kn = names.get(row.get("Dispatch_Id"), "?")
        per_kernel[kn] += 1
        if not kre.search(kn):
            continue
        if kept < 400000:
            w.writerow([kn] + [row.get(c, "") for c in cols]); kept += 1
        for c in by:
            if c in row:
                by[c][(kn, row[c])] += 1
        line = row.get("Instruction_Comment", "")
        if row.get("Wave_Issued_Instruction", "") in ("1", "true", "True"):
            issued_by_line[(kn, line)] += 1
        if "Stall_Reason" in row:
            stall_by_line[(kn, line)][row["Stall_Reason"]] += 1
    keep.close()
    print("# samples: %d; per kernel:" % tot)
    for k, v in per_kernel.most_common(25):
        print("   %8d  %5.1f %%  %s" % (v, 100.0 * v / max(tot, 1), k[:80]))
    nk = sum(v for (k, _), v in by["Instruction"].items())
    for c, top in (("Stall_Reason", 30), ("Instruction_Type", 30), ("Wave_Issued_Instruction", 10), ("Instruction_Comment", 120), ("Instruction", 150)):
        if not by[c]:
            continue
        print("\n## by %s (matching kernels: %d samples)" % (c, nk))
        for (kn, val), v in by[c].most_common(top):
            extra = ""
            if c == "Instruction_Comment" and stall_by_line:
                extra = "  issued %5.1f %%  stalls: %s" % (100.0 * issued_by_line[(kn, val)] / v, ", ".join("%s %d" % (a, b) for a, b in stall_by_line[(kn, val)].most_common(4)))
            print("   %8d  %5.2f %%  %-22s %s%s" % (v, 100.0 * v / max(nk, 1), kn[:22], val[-110:], extra))
