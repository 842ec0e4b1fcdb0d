import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# take the last 5 cnn_bwd_wino_k occurrences as step markers
idx=[i for i,r in enumerate(rows) if "cnn_bwd_wino_k" in r["Kernel_Name"]]
a,b=idx[-4],idx[-1]
span=int(rows[b]["Start_Timestamp"])-int(rows[a]["Start_Timestamp"])
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows[a:b])
print("steps",3,"span ms/step",span/3e6,"busy ms/step",busy/3e6,"idle frac",1-busy/span, "launches/step",(b-a)/3)
