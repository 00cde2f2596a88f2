"""TEST INFRASTRUCTURE ONLY — CPU restatement of the Silent-Speech NS decoder.

Nothing under this directory is part of the shipped product path.  It may be
imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline
leg, and only as the checker that HIP results are compared with.
"""
