import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "patch" not in k: continue
            k=k.replace("void (anonymous namespace)::","")[:44]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    m={c:sum(x)/len(x) for c,x in v.items()}
    for c in sorted(m): print("   %-30s %.4e"%(c,m[c]))
    if "SQ_INST_LEVEL_VMEM" in m and "SQ_INSTS_VMEM" in m: print("   avg VMEM latency (level/insts): %.0f"%(m["SQ_INST_LEVEL_VMEM"]/m["SQ_INSTS_VMEM"]))
    if "SQ_INST_LEVEL_LDS" in m and "SQ_INSTS_LDS" in m: print("   avg LDS latency (level/insts): %.0f"%(m["SQ_INST_LEVEL_LDS"]/m["SQ_INSTS_LDS"]))
