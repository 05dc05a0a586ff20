import sys, json
d = json.loads(sys.stdin.read())
r = d["roofline"]
print(sys.argv[1], "| ms/step %.4f | kern_ms %.4f | %.0f GB/s | frac %.3f | value %.3e" %
      (d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], d["value"]))
