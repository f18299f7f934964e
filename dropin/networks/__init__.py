"""Reference package name `networks` (dropin/README.md)."""
