"""phant_amd -- MI355X-native Keccak-256 / Merkle-Patricia hot path for phant.

Python mirror of the reference's module layout for this path only:
    phant_amd.crypto.hasher   <-> src/crypto/hasher.zig
    phant_amd.mpt             <-> src/mpt/mpt.zig  (+ the proof verifier it lacks)
    phant_amd.state           <-> the state-root surface of src/state
    phant_amd.engine_api      <-> the witness hook of src/engine_api/execution_payload.zig:175-178
    phant_amd.types, .signer  <-> the other bulk keccak256 users (receipt.zig logs bloom, Tx.hash, sender address)
Everything runs through the C-ABI in include/phant_gpu.h (libphant_gpu.so,
hand-written HIP for gfx950).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .context import Context, default_context  # noqa: F401
from .crypto import hasher  # noqa: F401
from . import mpt, state, witness, engine_api, types, signer, comm  # noqa: F401
