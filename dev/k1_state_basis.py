"""What the state basis of K1's block form (design.h: derive_block_form) costs in single precision, measured on a seeded capture
against the oracle's sequential scan: the block recurrence t_k = P t_{k-1} + taps . x, y_k = c . t_k + c2 xm, carried in fp32
(FMAs as the kernel issues them) in three bases - the recursion's own (v[n], v[n-1]), (value, slope), and the normal form of the
recursion matrix - against the same recurrence in double.  usage: python dev/k1_state_basis.py [golden case]"""
import os, sys, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests')
from oracle import pyoracle as po
import cases
f32=np.float32
def fma(a,b,c): return f32(np.float64(a)*np.float64(b)+np.float64(c))
def fmav(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(np.float32)

name=sys.argv[1] if len(sys.argv)>1 else 'config2_1s'
cfg,iq,bursts,gold=cases.load(name)
os_=cfg.oversample; freqs=list(cfg.freqs)
o=po.Oracle(cfg.centerfreq,freqs,oversample=os_)
nin=iq.size//2; D=nin//os_
tr=o.trace_all(D); o.process(iq.view(np.uint8),block_bytes=320000)
A,B=o.lpf()
A0,A1,A2=[float(a) for a in A]; B1,B2=float(B[1]),float(B[2])
x=(iq.reshape(-1,2).astype(np.float32)/f32(32768.0))
xc=(x[:,0].astype(np.float64)+1j*x[:,1].astype(np.float64))[:D*os_]
i=np.arange(257,dtype=np.float32); ang=(f32(2)*f32(np.pi)*(i%256)/f32(256)).astype(np.float32)
sin_l=np.sin(ang.astype(np.float64)).astype(np.float32); cos_l=np.cos(ang.astype(np.float64)).astype(np.float32)

hap=np.zeros(os_+2); hap[1]=1.0
for n in range(1,os_+1): hap[n+1]=B1*hap[n]+B2*(hap[n-1] if n>=2 else 0.0)
H=lambda n: hap[n+1]
g=np.array([[H(os_-1-j) for j in range(os_)],[H(os_-2-j) for j in range(os_)]])   # [2][os]
M=np.array([[B1,B2],[1.0,0.0]]); P=np.linalg.matrix_power(M,os_)
c=np.array([A0+A2/B2, A1-A2*B1/B2]); c2=-A2/B2
lam=np.roots([1,-B1,-B2])[0]; r=abs(lam); th=abs(np.angle(lam))
V=np.array([[r*np.cos(th), r*np.sin(th)],[1.0,0.0]])
bases={'current':np.eye(2), 'value/slope':np.array([[1.0,0.0],[1.0,-1.0]]), 'normal':np.linalg.inv(V)}
print(name,'poles r=%.5f theta=%.5f  P='%(r,th),P.round(3).tolist())
def run(T, xm32, exact=False):
    Ti=np.linalg.inv(T); g_=T@g; P_=T@P@Ti; c_=c@Ti
    dt=np.float64 if exact else np.float32
    g32=g_.astype(dt); P32=P_.astype(dt); c32=c_.astype(dt); c2_=dt(c2)
    mb=xm32.reshape(D,os_)
    # tap sums: FMA chain over j (fp32)
    acc=np.zeros((2,D),dtype=np.complex128)
    for k in range(2):
        ar=np.zeros(D,dtype=dt); ai=np.zeros(D,dtype=dt)
        for j in range(os_):
            if exact:
                ar=ar+g32[k,j]*mb[:,j].real; ai=ai+g32[k,j]*mb[:,j].imag
            else:
                ar=fmav(np.full(D,g32[k,j],dtype=np.float32), mb[:,j].real.astype(np.float32), ar)
                ai=fmav(np.full(D,g32[k,j],dtype=np.float32), mb[:,j].imag.astype(np.float32), ai)
        acc[k]=ar.astype(np.float64)+1j*ai.astype(np.float64)
    y=np.zeros(D,dtype=np.complex128)
    t0r=t0i=t1r=t1i=dt(0)
    ml=mb[:,-1]
    for k in range(D):
        a0r,a0i,a1r,a1i=dt(acc[0,k].real),dt(acc[0,k].imag),dt(acc[1,k].real),dt(acc[1,k].imag)
        if exact:
            n0r=P32[0,0]*t0r+P32[0,1]*t1r+a0r; n0i=P32[0,0]*t0i+P32[0,1]*t1i+a0i
            n1r=P32[1,0]*t0r+P32[1,1]*t1r+a1r; n1i=P32[1,0]*t0i+P32[1,1]*t1i+a1i
            yr=c32[0]*n0r+c32[1]*n1r+c2_*ml[k].real; yi=c32[0]*n0i+c32[1]*n1i+c2_*ml[k].imag
        else:
            n0r=fma(P32[0,0],t0r,fma(P32[0,1],t1r,a0r)); n0i=fma(P32[0,0],t0i,fma(P32[0,1],t1i,a0i))
            n1r=fma(P32[1,0],t0r,fma(P32[1,1],t1r,a1r)); n1i=fma(P32[1,0],t0i,fma(P32[1,1],t1i,a1i))
            yr=fma(c32[0],n0r,fma(c32[1],n1r,f32(c2_*f32(ml[k].real)))); yi=fma(c32[0],n0i,fma(c32[1],n1i,f32(c2_*f32(ml[k].imag))))
        t0r,t0i,t1r,t1i=n0r,n0i,n1r,n1i
        y[k]=complex(yr,yi)
    return y
for ch in ([0,3] if len(freqs)>3 else [0]):
    dphi=o.dphi(ch)
    ref=tr[ch,:D,0].astype(np.float64)+1j*tr[ch,:D,1].astype(np.float64); peak=np.abs(ref).max()
    n=np.arange(D*os_,dtype=np.uint64); ph=((n*np.uint64(dphi))&np.uint64(0xffffff)).astype(np.uint32)
    idx=(ph>>16).astype(np.int64); fr=(ph&0xffff).astype(np.float32)
    # the mixer as the device does it: (s,c)+(ds,dc)*F in fp32 (one FMA), then cos*x + sin*(ix)
    ds=((sin_l[idx+1]-sin_l[idx])*f32(1/65536.0)).astype(np.float32); dc=((cos_l[idx+1]-cos_l[idx])*f32(1/65536.0)).astype(np.float32)
    s=fmav(ds,fr,sin_l[idx]); cc=fmav(dc,fr,cos_l[idx])
    if dphi==0: s[:]=0; cc[:]=1
    xr=xc.real.astype(np.float32); xi=xc.imag.astype(np.float32)
    mr=fmav(cc,xr,(-(s*xi)).astype(np.float32)); mi=fmav(cc,xi,(s*xr).astype(np.float32))
    xm=(mr.astype(np.float64)+1j*mi.astype(np.float64))
    yex=run(np.eye(2),xm,exact=True)
    print(f' ch {ch}: oracle vs exact-arithmetic block form: {np.abs(ref-yex).max()/peak:.3e}; |v| scale: peak {peak:.4f}')
    for bn,T in bases.items():
        y=run(T,xm)
        print(f'   basis {bn:12s}: fp32 block form vs exact {np.abs(y-yex).max()/peak:.3e}   vs oracle {np.abs(y-ref).max()/peak:.3e}   rms vs oracle {np.sqrt(np.mean(np.abs(y-ref)**2))/peak:.3e}')
