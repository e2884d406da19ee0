"""frieda_amd — MI355X-native FRI-DAS commit / prove hot path behind frieda's commit()/generate_proof()/verify() API.

The package holds only what that path needs: `csrc/` (hand-written gfx950 kernels + the C-ABI host runtime,
built into `lib/libfrieda_hip.so`) and this thin host-side mirror of the reference interface.
"""
from .api import (  # noqa: F401
    BatchPipeline,
    Context,
    FriConfig,
    FriedaError,
    FriedaPanic,
    MultiContext,
    PcsConfig,
    Proof,
    ProofPipeline,
    batch_plan,
    commit,
    commit_and_generate_proof,
    default_context,
    generate_proof,
    verify,
    verify_samples,
    workspace_bytes,
)

__all__ = [
    "BatchPipeline",
    "Context",
    "FriConfig",
    "FriedaError",
    "FriedaPanic",
    "MultiContext",
    "PcsConfig",
    "Proof",
    "ProofPipeline",
    "batch_plan",
    "commit",
    "commit_and_generate_proof",
    "default_context",
    "generate_proof",
    "verify",
    "verify_samples",
    "workspace_bytes",
]
