"""drprg_amd -- MI355X-native predict hot path of drprg."""
from .pandora import Context, DependencyError, Pandora  # noqa: F401
