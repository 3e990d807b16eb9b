"""Developer tooling that ships with the package because GPU tests use it (nothing here is on the product path)."""
