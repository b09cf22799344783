import csv,glob,collections,sys
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True))[-1]
steps=float(sys.argv[2]) if len(sys.argv)>2 else 7
pat=sys.argv[3] if len(sys.argv)>3 else 'gemm'
rows=list(csv.DictReader(open(f)))
agg=collections.OrderedDict()
for x in rows:
    n=x['Kernel_Name']
    if not any(p in n for p in pat.split(',')): continue
    key=(n[:34],int(x['Grid_Size_X'])//int(x['Workgroup_Size_X']),x['Grid_Size_Y'],x['Grid_Size_Z'])
    d=(int(x['End_Timestamp'])-int(x['Start_Timestamp']))/1e3
    agg.setdefault(key,[]).append(d)
for k,v in agg.items():
    print(k, 'n/step=%.1f avg=%.1fus total/step=%.2fms'%(len(v)/steps,sum(v)/len(v),sum(v)/steps/1e3))
