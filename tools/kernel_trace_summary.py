import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'group_norm' in n or 'relu_mask' in n or 'dropout' in n or 'add_kernel' in n or 'sum_kernel' in n or 'axpy' in n:
        d[(n.split('(')[0][-40:], r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    v.sort(); print(f"{k[0]:42s} grid {k[1]:>8s} n {len(v):4d} median {v[len(v)//2]:7.2f} us")
