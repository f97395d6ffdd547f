"""DEV TOOL (GPU): frames per second as a function of the call size -- the reference's own call pattern (1152 samples per
encodeBuffer call), 8 and 64 frames per call, mono and stereo; the record is profiles/r02_call_latency.txt."""
import sys,time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import lamejs_amd, pcm, numpy as np
for ch in (1,2):
    L,R=pcm.sine(1152*400,ch)
    for chunk in (1152, 1152*8, 1152*64):
        enc=lamejs_amd.Mp3Encoder(ch,44100,128)
        enc.encodeBuffer(L[:chunk*2], None if R is None else R[:chunk*2])
        t0=time.perf_counter(); n=0
        for p in range(chunk*2, len(L)-chunk+1, chunk):
            enc.encodeBuffer(L[p:p+chunk], None if R is None else R[p:p+chunk]); n+=1
        dt=time.perf_counter()-t0
        print(f"ch={ch} chunk={chunk//1152} frames: {1e6*dt/n:.0f} us/call, {n*(chunk//1152)/dt:.0f} frames/s")
