from timediff_amd.blocks import RNN, EncoderDecoderRNN, NeuralCDE, RNNBlock, SinusoidalPosEmb  # noqa: F401
