"""Reference package name `modules` (dropin/README.md)."""
