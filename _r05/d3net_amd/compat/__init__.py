"""Reference-side bindings: modules that take the place of the reference's own import names."""
