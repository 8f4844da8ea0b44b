import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["whole_update"])
tot=0
for k,v in d["kernels"].items():
    print("%-24s %8.4f ms %7.2f TF"%(k,v["avg_ms"],v["tflops"])); tot+=v["avg_ms"]
print("sum per minibatch step: %.4f ms"%tot)
