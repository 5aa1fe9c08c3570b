import json
r={v:json.loads(open("gpurun_out/abl/abl%s.json"%v).read().strip().splitlines()[-1]) for v in ("_tune","_noA","_noB")}
for k in sorted(r["_tune"]["kernels"]):
    t=[r[v]["kernels"][k]["ms_per_step"] for v in ("_tune","_noA","_noB")]
    print("%-18s %.3f  noA %.3f (%+.1f%%)  noB %.3f (%+.1f%%)"%(k,t[0],t[1],100*(t[1]/t[0]-1),t[2],100*(t[2]/t[0]-1)))
print("step", [round(r[v]["ms_per_step"],2) for v in r])
