import csv,sys,glob
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True))[-1]
steps=float(sys.argv[2]) if len(sys.argv)>2 else 7
r=list(csv.DictReader(open(f)))
tot=sum(float(x['TotalDurationNs']) for x in r)
for x in r[:int(sys.argv[3]) if len(sys.argv)>3 else 16]:
    print('%-78s calls/step=%6.1f ms/step=%8.3f avg=%9.1fus %5.1f%%'%(x['Name'][:78],float(x['Calls'])/steps,float(x['TotalDurationNs'])/1e6/steps,float(x['AverageNs'])/1e3,100*float(x['TotalDurationNs'])/tot))
print('total GPU ms/step',tot/1e6/steps)
