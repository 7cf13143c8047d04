"""sgl_amd.operators -- MI355X-native GraphOp / MessageOp plugins (API of sgl/operators)."""
