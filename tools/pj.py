"""One-line digest of a bench.py JSON line read from stdin (diagnostic helper for gpurun command lines)."""
import json, sys
j = json.loads(sys.stdin.read()); k = j["roofline"]["kernel_span_ms_each"]; s = j["roofline"].get("serial_steps") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", round(j["value"]), round(j["ms_per_step"], 1), "span max", round(max(k)), "serial", round(s.get("ms_per_step", 0)))
