import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ours=[r for r in rows if r['Kernel_Name'].startswith(('beam_step','void masked_lsm','plan_','beam_init','void sid','group_adv','state_'))]
ours.sort(key=lambda r:int(r['Start_Timestamp']))
seq=ours[-int(sys.argv[2]):]
t0=int(seq[0]['Start_Timestamp']); prev=None
for r in seq:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:28]:28s} start {(s-t0)/1e3:9.1f} dur {(e-s)/1e3:7.1f} gap {((s-prev)/1e3 if prev else 0):6.1f}")
    prev=e
