"""Conv blocks: plain ConvBlock and the V1 / private passport layers."""
