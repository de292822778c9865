"""The pieces of bench.py (the driver's entry point at the repository root keeps its CLI and its one JSON line):
  common.py        constants of the roofline objects, the synthetic inputs, the host's CPU budget
  launch.py        starting the ranks, the ready / report exchanges that keep one failing rank from costing the others
  headline.py      main(): the MSM steps, the timed region, the JSON line
  extras.py        configs C2 .. C5 as `extra` entries
  cpu_baseline.py  the oracle (and the reference's own algorithm) timed on the host cores"""
