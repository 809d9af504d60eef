#!/usr/bin/env python3
"""Quality check of the table hash (shk::mix_key in sharkmer_amd/csrc/shk_device.hip.h): page
occupancy spread and home-bucket overflow against a Poisson process, plus the bijection property,
on random, AT-rich, tandem-repeat and sequential keys.  CPU only (numpy restatement of the device
formula: ONE multiplication by an odd constant mod 2^2k — a 32-bit constant up to 42 key bits, a 64-bit one beyond)."""
import numpy as np
M32=np.uint64(0xC2B2AE35); M64=np.uint64(0x9E3779B97F4A7C15)
def mix(x,bits):
    mask=np.uint64((1<<bits)-1)
    return (x*(M32 if bits<=42 else M64))&mask
def canon(codes,k):
    n=len(codes)-k+1
    f=np.zeros(n,dtype=np.uint64); r=np.zeros(n,dtype=np.uint64)
    for j in range(k):
        b=codes[j:j+n].astype(np.uint64)
        f=(f<<np.uint64(2))|b
        r|=(np.uint64(3)-b)<<np.uint64(2*j)
    return np.unique(np.minimum(f,r))
def stats(keys,name,k,lp=10):
    bits=2*k
    y=mix(keys,bits)
    assert len(np.unique(y))==len(keys)
    Y=y<<np.uint64(64-bits)
    page=(Y>>np.uint64(64-lp)).astype(np.int64)
    bucket=((Y>>np.uint64(64-lp-11))&np.uint64(2047)).astype(np.int64)
    cnt=np.bincount(page,minlength=1<<lp)
    lam=len(keys)/(1<<lp)
    pb=np.bincount(page*2048+bucket,minlength=(1<<lp)*2048)
    lb=len(keys)/((1<<lp)*2048)
    # poisson expectation of P(bucket load>4)
    from math import exp,factorial
    p_over=1-sum(exp(-lb)*lb**i/factorial(i) for i in range(5))
    print(f"{name:12s} k={k} n={len(keys):8d} page mean {lam:7.0f} max {cnt.max():6d} std {cnt.std():6.1f} (poisson {lam**0.5:5.1f}) | bucket load {lb:.2f} frac>4 {np.mean(pb>4):.4f} (poisson {p_over:.4f}) max {pb.max()}")
rng=np.random.default_rng(1)
stats(canon(rng.integers(0,4,size=3_000_000),21),"random",21)
unit=rng.integers(0,4,size=7); g2=np.tile(unit,300000); mut=rng.random(len(g2))<0.02; g2[mut]=rng.integers(0,4,size=mut.sum())
stats(canon(g2,21),"tandem7",21)
stats(canon(rng.choice(4,size=2_000_000,p=[0.45,0.05,0.05,0.45]),21),"AT-rich",21)
stats(np.arange(3_000_000,dtype=np.uint64)*np.uint64(4),"sequential",21)
stats(canon(rng.integers(0,4,size=2_000_000),31),"random",31)
stats(np.arange(3_000_000,dtype=np.uint64)*np.uint64(4),"sequential",31)
stats(canon(rng.integers(0,4,size=2_000_000),13),"random",13,lp=8)
stats(np.arange(1<<18,dtype=np.uint64),"all k=9",9,lp=2)
