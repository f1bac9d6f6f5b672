"""Development aid: registers, scratch and code size of every compiled PDIP instance, read off the objects build.py left in csrc/_obj
(llvm-readelf --notes of the unbundled gfx950 code object).  Run after a build: a kernel change that makes a headline instance spill shows here."""
import glob,os,subprocess,re,sys,tempfile
LL="/opt/rocm/lib/llvm/bin"
objs=sorted(glob.glob("/root/repo/lsc_dr_planner_amd/csrc/_obj%s/inst_*.o" % (("_" + os.environ["LSCQP_AB"]) if os.environ.get("LSCQP_AB") else "")))
out=[]
for o in objs:
    with tempfile.TemporaryDirectory() as td:
        fat=os.path.join(td,'f'); co=os.path.join(td,'c')
        subprocess.check_call([LL+'/llvm-objcopy','-O','binary','--only-section=.hip_fatbin',o,fat])
        subprocess.check_call([LL+'/clang-offload-bundler','--unbundle','--type=o','--input='+fat,'--targets=hipv4-amdgcn-amd-amdhsa--gfx950','--output='+co],stderr=subprocess.DEVNULL)
        t=subprocess.check_output([LL+'/llvm-readelf','--notes',co],text=True)
        m=lambda k: re.search(r'\.%s:\s+(\d+)'%k,t)
        vals={k:int(m(k).group(1)) for k in ('private_segment_fixed_size','vgpr_count','agpr_count','sgpr_count','group_segment_fixed_size') if m(k)}
        sz=os.path.getsize(co)
        out.append((os.path.basename(o),vals,sz))
for n,v,sz in out:
    print(n, 'scratch',v.get('private_segment_fixed_size'),'vgpr',v.get('vgpr_count'),'agpr',v.get('agpr_count'),'co_bytes',sz)
