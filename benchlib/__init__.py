"""bench.py's parts: workload (flags, data, step driver), launch (N ranks, watchdog), probes (roofline), evalbench,
baseline (CPU leg), dropin_bench, report (the JSON line)."""
