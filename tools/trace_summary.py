"""prints the per-launch durations of the NTT kernels from a rocprofv3 --kernel-trace csv"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "ntt_pass" in n or "lde_mid" in n or "lde_coset" in n:
        print("%-70s %8.3f ms grid %s wg %sx%s lds %s" % (n.replace("(anonymous namespace)::", "")[:70], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
              r.get("Grid_Size_X"), r.get("Workgroup_Size_X"), r.get("Workgroup_Size_Y"), r.get("LDS_Block_Size")))
