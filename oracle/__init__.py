"""CPU oracle for the V-cycle hot path -- TEST INFRASTRUCTURE ONLY.

Import rule (enforced by tests/test_abi.py::test_product_never_imports_the_oracle): only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import `oracle`; the product
package `learnmultigrid_amd` never does and fails loudly without its HIP library.
"""
