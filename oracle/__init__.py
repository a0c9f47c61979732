"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's self-play hot path (rules, search, net),
used as the checker by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under ataxxzero_amd/ imports this package.
"""
