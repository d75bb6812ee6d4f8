"""cogstream_amd -- MI355X-native CogReasoner streaming-VQA hot path (HIP kernels behind a C ABI).

Importing the package does not touch the GPU; importing `cogstream_amd._lib` (done by every compute
module) requires the built libcogs_hip.so and raises otherwise -- there is no CPU fallback."""

__version__ = "0.1.0"
