"""Per-group begin / end times of a decode, from the marker records of a diagnostic build:
  tools/build_variant.sh grouptimes -DFSMC_DIAG_GROUP_TIMES
  FSMC_HIP_LIB=fastsmc_amd/variants/libgrouptimes.so python bench.py --workload c3 --pairs 262144 --dump-records r.npy
  python tools/analyse_group_times.py r.npy
Prints the spread of the groups' durations, when they ended, how many groups each resident wave took, per XCD."""
import numpy as np, sys
r=np.load(sys.argv[1])
m=r[r["start"]==-1]
print("markers", m.size, "records", r.size)
t0=m["post_mean"].astype(np.float64); t1=m["prob"].astype(np.float64)
base=t0.min(); t0-=base; t1-=base
dur=t1-t0
hw=m["map"].view(np.int32)
xcc=(hw>>16)&0xF; cu=(hw>>8)&0xF; simd=(hw>>4)&0x3; se=(hw>>13)&0x7; sh=(hw>>12)&1
print("kernel span ms", t1.max())
print("dur ms: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f"%(dur.min(),np.percentile(dur,10),np.median(dur),np.percentile(dur,90),dur.max()))
# by start order (first group of a slot vs second)
first = t0 < 50
print("first-round groups:", first.sum(), "dur median %.1f max %.1f"%(np.median(dur[first]), dur[first].max()))
print("later groups:", (~first).sum(), "dur median %.1f min %.1f max %.1f"%(np.median(dur[~first]), dur[~first].min(), dur[~first].max()))
for x in range(8):
    s=xcc==x
    if s.sum(): print("xcc",x,"n",s.sum(),"median dur %.1f"%np.median(dur[s]), "first-round median %.1f"%np.median(dur[s&first]) if (s&first).sum() else "")
# histogram of end times
h,e=np.histogram(t1,bins=12); print("end-time histogram", list(zip(e[:-1].astype(int),h)))
h,e=np.histogram(dur,bins=10); print("duration histogram", list(zip(e[:-1].astype(int),h)))
# per slot: groups taken
b=m["end"]; cnt=np.bincount(b, minlength=2048); print("groups per slot: ", np.bincount(cnt))
