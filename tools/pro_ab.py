import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import conv_bench as cb
S = ["fwd:4,32,57,256,256,3,1", "fwd:4,32,57,256,1024,1,1", "fwd:4,32,57,1024,256,1,1", "fwd:4,128,228,64,64,3,1", "fwd:4,64,114,128,128,3,1", "wgrad:4,32,57,256,256,3,1", "wgrad:4,32,57,1024,256,1,1"]
for pro in (True, False, True, False):
    print("PRO", pro)
    for s in S: cb.run(s, pro=pro, stats=True)
