"""lens_trace_amd -- MI355X-native ray-trace backend behind lens_trace's Renderer plugin surface."""
