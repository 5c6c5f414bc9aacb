"""Stand-ins for third-party packages the reference's harness imports but this environment lacks."""
