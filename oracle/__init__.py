"""CPU oracle for the FA2 forward path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (flash_attention_from_scratch_amd) never imports this.
"""
