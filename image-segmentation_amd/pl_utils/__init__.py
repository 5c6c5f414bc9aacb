"""Lightning-style wrappers of the reference's harness (`pl_utils/`), on the real pytorch_lightning when it is
installed and on segk_compat.lightning otherwise."""
from segk_compat.lightning import install as _install

_install()
