"""phant_amd.types <-> the keccak256 users of src/types/ (receipt.zig logs bloom, transaction.zig hashes)."""
from . import receipt, transaction  # noqa: F401
