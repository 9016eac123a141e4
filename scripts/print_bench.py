import json
import sys
d = json.loads(sys.stdin.read())
print(sys.argv[1], round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms")
