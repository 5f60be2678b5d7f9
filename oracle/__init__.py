"""CPU oracle for the mixture-of-tokenizers embedding front-end -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg.
The product package (mixture-of-tokenizers_amd/) never imports this.
"""
