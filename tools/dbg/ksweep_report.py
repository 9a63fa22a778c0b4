import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'gemm_glds' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
REP = 12
i = 0
for (M, N) in ((8000, 2048), (8000, 512)):
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        d = sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[i:i+REP]); i += REP
        med = d[len(d)//2] / 1e3
        print(f"M {M} N {N} K {K:5d}: {med:7.1f} us  {2.0*M*N*K/med/1e6:7.1f} TF")
