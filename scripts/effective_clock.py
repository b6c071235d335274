"""Effective shader clock per kernel from a `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- ...` run:
GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel wall time (MI355X_MICROARCH.md, DVFS give-back; reads high on dispatches
shorter than ~0.3 ms).  usage: effective_clock.py DIR [name-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or pat not in r["Kernel_Name"] or r["Dispatch_Id"] not in dur: continue
    a = acc[r["Kernel_Name"][:70]]; a[0] += float(r["Counter_Value"]) / 8; a[1] += dur[r["Dispatch_Id"]]; a[2] += 1
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-72s launches %5d  avg %8.1f us  effective clock %.2f GHz" % (k, n, ns / n / 1e3, cyc / ns))
