"""Development aid: randomized sweep of the voxel map and the three corridor constructors against the CPU oracle, bit for bit:
12 random worlds (2-D / 3-D, bounds off the grid, resolutions 0.05 / 0.1 / 0.2, radii 0.1-0.25) x 200 agents.  Needs a GPU."""
import numpy as np, sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsc_dr_planner_amd import api
from oracle import oracle as O
dev=torch.device("cuda",0)
up=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
nbad=0; tot=0
SEEDS = range(*[int(v) for v in sys.argv[sys.argv.index("--seeds") + 1:sys.argv.index("--seeds") + 3]]) if "--seeds" in sys.argv else range(12)
for seed in SEEDS:
    rng=np.random.default_rng(seed)
    dim=3 if seed%2 else 2
    wmin=np.array([-6.0,-6.0,0.0])+rng.choice([0,0.03,-0.27],3); wmax=np.array([6.0,6.0,3.0 if dim==3 else 2.5])+rng.choice([0,0.04,0.31],3)
    nb=rng.integers(40,260)
    c=rng.uniform(wmin,wmax,(nb,3)); s=rng.choice([0.3,0.5,0.8,1.2],(nb,3))
    if dim==2: c[:,2]=1.25; s[:,2]=2.5
    boxes=np.concatenate([c,s],axis=1)
    res=float(rng.choice([0.1,0.1,0.2,0.05]))
    om=O.Map(boxes,wmin,wmax,res,1.0); gm=api.WorldMap(boxes,wmin,wmax,res,1.0)
    if "--prepare" in sys.argv: gm.prepare(0.25 if seed%3 else 0.15)  # lscqp_map_prepare: the free-space table (0.15: the larger agents stay on the exact path)
    occ,near=gm.download()
    assert np.array_equal(occ,om.occ()) and np.array_equal(near,om.nearest()),("map",seed)
    n=200; M=5
    starts=np.float32(rng.uniform(wmin+0.3,wmax-0.3,(n,3))).astype(np.float64)
    if dim==2: starts[:,2]=0.6
    radius=rng.choice([0.15,0.2,0.25,0.1],n)
    sol=api.Solver(api.make_desc(M=M,dim=dim,world_min=wmin,world_max=wmax))
    def gpu(mode,P,sfc):
        d_sfc=torch.from_numpy(sfc.view(np.float64).reshape(-1).copy()).to(dev); d_st=torch.full((n,),-7,dtype=torch.int32,device=dev)
        sol.construct_sfc_device(gm,mode,n,up(P.reshape(-1)),up(radius),d_sfc,d_st); torch.cuda.synchronize()
        return d_sfc.cpu().numpy().view(api.BOX_DTYPE).reshape(n,M),d_st.cpu().numpy()
    P0=np.repeat(starts[:,None,:],3,axis=1)
    want=np.zeros((n,M),O.BOX_DTYPE); stw=om.construct_sfc(0,P0,radius,want)
    got,stg=gpu(0,P0,np.zeros((n,M),api.BOX_DTYPE))
    ok=stw==1
    bad=(not np.array_equal(stg,stw)) or (not np.array_equal(got["bmin"][ok],want["bmin"][ok])) or (not np.array_equal(got["bmax"][ok],want["bmax"][ok]))
    nbad+=bad; tot+=1
    base=want.copy(); base[~ok]=base[np.nonzero(ok)[0][0]]
    d=rng.normal(size=(n,3)); 
    if dim==2: d[:,2]=0
    d/=np.linalg.norm(d,axis=1,keepdims=True)
    last=np.float32(starts+rng.uniform(0.05,0.6)*d).astype(np.float64); goal=np.float32(starts+rng.uniform(0.3,1.2)*d).astype(np.float64); wp=np.float32(starts+0.5*d).astype(np.float64)
    P=np.stack([last,goal,wp],axis=1)
    for mode in (1,2):
        w=base.copy(); stw=om.construct_sfc(mode,P,radius,w)
        got,stg=gpu(mode,P,base.copy())
        bad=(not np.array_equal(stg,stw)) or (not np.array_equal(got["bmin"],w["bmin"])) or (not np.array_equal(got["bmax"],w["bmax"]))
        if bad: print("BAD seed",seed,"mode",mode,"res",res,(stg!=stw).sum())
        nbad+=bad; tot+=1
    gm.close()
    print("seed",seed,"dim",dim,"res",res,"boxes",nb,"init ok",int(ok.sum()),"done"); sys.stdout.flush()
print("bad",nbad,"of",tot)
