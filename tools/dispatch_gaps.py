import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
prev=None
for r in rows[-12:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(r["Kernel_Name"][:44].ljust(44), "dur %.2f us"%((e-s)/1000), "gap", (s-prev)/1000 if prev else None)
    prev=e
