"""bulletproofs_amd -- MI355X-native MSM + inner-product-argument engine behind the
call surface of wborgeaud/python-bulletproofs (src/pippenger, src/innerproduct and
their callers).  See DESIGN.md."""
__version__ = "0.1.0"
