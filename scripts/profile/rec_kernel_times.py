import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
# group consecutive launches by (kernel, grid, wg)
from collections import OrderedDict
agg=OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if 'k_rec_e2r' in n or 'k_rec_r2e' in n:
        key=(n.split('(')[0].replace('void pxm::',''), int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), r['Workgroup_Size_X'])
        agg.setdefault(key,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    v=sorted(v)
    print(k[0].ljust(18),'wgs',str(k[1]).rjust(5),'threads',k[2].rjust(4),'n',len(v),f'median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}')
