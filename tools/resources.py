import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g, subprocess, os, json, sys
srcs=os.path.join(g.CSRC,"pagk_hip.hip")
r=subprocess.run(["/opt/rocm/bin/hipcc",*g.HIPCC_FLAGS,*sys.argv[1:],"-Rpass-analysis=kernel-resource-usage","-o","/tmp/t.so",srcs],capture_output=True,text=True)
res=g._parse_resource_remarks(r.stderr)
print(r.stderr[-3000:] if r.returncode else "ok")
for k,v in res.items():
    if ('track_block' in k or 'resume' in k) and 'Li2E' in k: print(k[:60], v['vgprs'], v['vgpr_spills'], v['sgpr_spills'])
