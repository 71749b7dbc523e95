"""MI355X-native DMLNet hot path.  Put this directory on sys.path and `import network`,
`import utils` exactly as the reference's DeepLabV3Plus-Pytorch/ directory is used."""
