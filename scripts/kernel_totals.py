import csv,sys,glob,collections
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*_kernel_trace.csv',recursive=True)[0]
    agg=collections.defaultdict(lambda:[0,0])
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']; t=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
        agg[n][0]+=t; agg[n][1]+=1
    print(d)
    tot=sum(v[0] for v in agg.values())
    for n,(t,c) in sorted(agg.items(), key=lambda x:-x[1][0])[:14]:
        print(f"  {t/5/1000:9.1f} us/step  {c/5:5.1f} x  {n[:150]}")
    print('  total/step', tot/5/1000)
